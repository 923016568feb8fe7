#!/bin/bash
# First thing to run on a node with more than one MI355X: everything of this repository that needs N > 1 devices and has never
# executed (the build box and the gpurun boxes expose one GPU).  One command, one log:
#     bash tools/scale_check.sh [max_gpus] 2>&1 | tee gpurun_out/scale_check.log
# 1. the four world-2 RCCL tests (sharded objects = single process, bench self-launch, data-parallel training step with the flat
#    averager and with DistributedDataParallel);
# 2. bench.py at N = 1, 2, 4, 8 (as far as devices exist): per-rank min / max / sum of images/s from the line's own fields, and the
#    weak-scaling efficiency the driver would compute from the per-N values.
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
