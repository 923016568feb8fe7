# r05: 128 x 224 ring tile (config 27): bit-identity with the other tiles, conv parity, micro-bench, headline A/B
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5p
(timeout 300 python tools/tile_invariance_check.py 2>&1 | tail -12) | tee gpurun_out/r5p/tile_invariance.txt
(timeout 600 python -m pytest tests/test_kernels_gpu.py -q --tb=short -p no:cacheprovider -x -k "tile_shape or conv_bf16x3" 2>&1 | tail -8) | tee gpurun_out/r5p/tests1.txt
for only in "l1_out2a 3x3" "l1_outconv" "layer2 3x3 196" "layer2.0" "l2_out2b"; do
  timeout 200 python tools/conv_bench.py --only "$only" --cfgs 22,25,27,-1 --iters 30 2>/dev/null | grep -v "^/" | tee -a gpurun_out/r5p/conv_bench_224.txt
done
run() { local label=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --steps 30 --warmup 4 --cpu-seconds 0 --no-roofline --no-legs "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', d['value'], d['ms_per_image'])" | tee -a gpurun_out/r5p/ab_224.txt
}
for rep in 1 2 3; do
  run tile224_on X=1 --
  run tile224_off OPP_TILE_224=0 --
done
run s1_on X=1 -- --streams 1
run s1_off OPP_TILE_224=0 -- --streams 1
