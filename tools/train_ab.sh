# training-step timing A/B of one environment switch (bench.py train_leg):  bash tools/train_ab.sh OPP_TRAIN_CHANNELS_LAST 0 1
VAR=$1; shift
python - "$VAR" "$@" <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import torch
import bench
dev = torch.device("cuda", 0)
var, vals = sys.argv[1], sys.argv[2:]
for rep in range(2):
    for v in vals:
        os.environ[var] = v
        r = bench.train_leg(torch, dev, "bf16x3", nsteps=3)
        print("%s=%s" % (var, v), {k: r[k] for k in ("forward_ms", "step_ms", "loss")}, flush=True)
PY
