#!/bin/bash
# 128 x 224 ring tile (gemm_mfma.hip config 27) against 128 x 256 / 128 x 128 on the 196(->224)-column layers, one box, one gpurun call:
# bit-identity and parity tests, the micro-bench, the headline A/B (four and one forward in flight), counters of the half-round layer.
#     bash tools/ab_tile224.sh > gpurun_out/ab_tile224.log 2>&1         (numbers behind profiles/r05_conv_bench_224_columns.txt)
set -x
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/ab_tile224; mkdir -p $OUT
(timeout 300 python tools/tile_invariance_check.py 2>&1 | tail -6) | tee $OUT/tile_invariance.txt
(timeout 600 python -m pytest tests/test_kernels_gpu.py -q --tb=short -p no:cacheprovider -x -k "tile_shape or conv_bf16x3 or linear_bf16x3" 2>&1 | tail -5) | tee $OUT/tests.txt
for only in "l1_out2a 3x3" "l1_outconv" "layer2 3x3 196" "layer2.0" "l2_out2b"; do
  timeout 200 python tools/conv_bench.py --only "$only" --cfgs 22,25,27 --iters 30 2>/dev/null | grep -v "^/" | tee -a $OUT/conv_bench_224.txt
done
run() { local label=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --steps 30 --warmup 4 --cpu-seconds 0 --no-roofline --no-legs "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', d['value'], d['ms_per_image'])" | tee -a $OUT/ab_224.txt
}
for rep in 1 2; do
  run tile224_off OPP_TILE_224=0 --
  run tile224_all OPP_TILE_224=2 --
done
run s1_off OPP_TILE_224=0 -- --streams 1
run s1_default X=1 -- --streams 1
bash tools/pmc_conv.sh "layer2 3x3 196" $OUT/pmc128 --cfgs 22,27 > /dev/null 2>&1
python tools/pmc_conv_summary.py $OUT/pmc128 > $OUT/pmc128.txt
find $OUT/pmc128 -name "*.csv" -delete
