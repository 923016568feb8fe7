# training-step timing with / without the HIP Linear backward (bench.py train_leg), torch.profiler table of the step
python - <<'PY'
import sys, os, json, time
sys.path.insert(0, os.getcwd())
import torch
import bench
dev = torch.device("cuda", 0)
for flag in ("1", "0"):
    os.environ["OPP_TRAIN_HIP_LINEAR"] = flag
    r = bench.train_leg(torch, dev, "bf16x3", nsteps=3)
    print("OPP_TRAIN_HIP_LINEAR=%s" % flag, {k: r[k] for k in ("forward_ms", "step_ms", "loss")})
PY
