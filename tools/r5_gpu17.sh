set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5q
run() { local label=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --steps 30 --warmup 4 --cpu-seconds 0 --no-roofline --no-legs "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', d['value'], d['ms_per_image'])" | tee -a gpurun_out/r5q/ab_224.txt
}
for rep in 1 2 3; do
  run tile224_on X=1 --
  run tile224_off OPP_TILE_224=0 --
done
run s1_on X=1 -- --streams 1
run s1_off OPP_TILE_224=0 -- --streams 1
python tools/matcher_bench.py --reps 50 2>/dev/null | head -1 | tee gpurun_out/r5q/matcher_bench.txt
python - <<'PY' 2>&1 | tail -5 | tee gpurun_out/r5q/train_ab.txt
import os, sys, torch
sys.path.insert(0, '.')
import bench
r = bench.train_leg(torch, torch.device("cuda:0"), "bf16x3", nsteps=6)
print("train step ms (224 on)", r.get("step_ms"), r.get("forward_ms"))
PY
OPP_TILE_224=0 python - <<'PY' 2>&1 | tail -5 | tee -a gpurun_out/r5q/train_ab.txt
import os, sys, torch
sys.path.insert(0, '.')
import bench
r = bench.train_leg(torch, torch.device("cuda:0"), "bf16x3", nsteps=6)
print("train step ms (224 off)", r.get("step_ms"), r.get("forward_ms"))
PY
