#!/bin/bash
# First thing to run on a node with more than one MI355X: everything of this repository that needs N > 1 devices and has never
# executed (the build box and the gpurun boxes expose one GPU).  One command, one log:
#     bash tools/scale_check.sh [max_gpus] 2>&1 | tee gpurun_out/scale_check.log
# 1. the four world-2 RCCL tests (sharded objects = single process, bench self-launch, data-parallel training step with the flat
#    averager and with DistributedDataParallel);
# 2. bench.py at N = 1, 2, 4, 8 (as far as devices exist): per-rank min / max / sum of images/s from the line's own fields, and the
#    weak-scaling efficiency the driver would compute from the per-N values -- the headline (BASELINE configs[1]) AND configs[3]: one object
#    per GPU, full coarse-to-fine, 15 000-point clouds (bench.py --fine --n-points 15000);
# 3. configs[4]: the data-parallel training step at B = 4 x 15 000 points per GPU, ONE flat all-reduce per step (tools/train_scale.py).
set -u
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
NG=$(python -c "import torch; print(torch.cuda.device_count())")
MAXG=${1:-$NG}
echo "== visible GPUs: $NG (running up to $MAXG)"
if [ "$NG" -ge 2 ]; then
  echo "== world-2 RCCL tests"
  python -m pytest tests/test_multi_gpu.py -q -m gpu --tb=short -p no:cacheprovider
else
  echo "== fewer than 2 GPUs: the world-2 tests would skip"
fi
mkdir -p gpurun_out
BASE=""
for N in 1 2 4 8; do
  [ "$N" -le "$MAXG" ] && [ "$N" -le "$NG" ] || continue
  echo "== bench.py --gpus $N"
  python bench.py --gpus "$N" --steps 20 --warmup 5 --no-legs 2>/dev/null | tail -1 > "gpurun_out/scale_n$N.json" || { echo "bench failed at N=$N"; continue; }
  python - "$N" "gpurun_out/scale_n$N.json" "$BASE" <<'PY'
import json, sys
n, path, base = int(sys.argv[1]), sys.argv[2], sys.argv[3]
line = json.loads(open(path).read())
cfg = line.get("config", {})
pr = cfg.get("per_rank_images_per_s", {})
print("N=%d  value %.1f %s  ranks seen %s  per-rank min/max/sum %s/%s/%s" % (
    n, line["value"], line["unit"], cfg.get("n_ranks_seen"), pr.get("min"), pr.get("max"), pr.get("sum")))
if base:
    b = float(base)
    print("     weak-scaling efficiency vs N=1: %.3f" % (line["value"] / (n * b)))
PY
  if [ "$N" -eq 1 ]; then BASE=$(python -c "import json;print(json.loads(open('gpurun_out/scale_n1.json').read())['value'])"); fi
done

echo "== configs[3]: one object per GPU, full coarse-to-fine, 15 000 points"
BASE3=""
for N in 1 2 4 8; do
  [ "$N" -le "$MAXG" ] && [ "$N" -le "$NG" ] || continue
  python bench.py --gpus "$N" --fine --thr 0 --n-points 15000 --steps 10 --warmup 3 --no-legs --no-roofline --cpu-seconds 0 2>/dev/null | tail -1 > "gpurun_out/scale_cfg3_n$N.json" || { echo "configs[3] bench failed at N=$N"; continue; }
  V=$(python -c "import json;print(json.loads(open('gpurun_out/scale_cfg3_n$N.json').read())['value'])")
  [ "$N" -eq 1 ] && BASE3=$V
  python -c "print('N=$N  %.1f images/s  efficiency vs N=1: %.3f' % ($V, $V / ($N * $BASE3)))"
done
echo "== configs[4]: data-parallel training step, B = 4 x 15 000 points per GPU"
BASE4=""
for N in 1 2 4 8; do
  [ "$N" -le "$MAXG" ] && [ "$N" -le "$NG" ] || continue
  python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port $((29600 + N)) tools/train_scale.py 2>/dev/null | tail -1 > "gpurun_out/scale_cfg4_n$N.json" || { echo "configs[4] failed at N=$N"; continue; }
  cat "gpurun_out/scale_cfg4_n$N.json"
  V=$(python -c "import json;print(json.loads(open('gpurun_out/scale_cfg4_n$N.json').read())['value'])")
  [ "$N" -eq 1 ] && BASE4=$V
  python -c "print('N=$N  %.2f samples/s  efficiency vs N=1: %.3f' % ($V, $V / ($N * $BASE4)))"
done
