set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5r
(timeout 300 python tools/tile_invariance_check.py 2>&1 | tail -6) | tee gpurun_out/r5r/tile_invariance.txt
(timeout 600 python -m pytest tests/test_kernels_gpu.py -q --tb=short -p no:cacheprovider -x -k "tile_shape or conv_bf16x3" 2>&1 | tail -5) | tee gpurun_out/r5r/tests1.txt
for only in "l1_out2a 3x3" "l1_outconv" "layer2 3x3 196" "layer2.0" "l2_out2b"; do
  timeout 200 python tools/conv_bench.py --only "$only" --cfgs 22,25,27 --iters 30 2>/dev/null | grep -v "^/" | tee -a gpurun_out/r5r/conv_bench_224.txt
done
run() { local label=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --steps 30 --warmup 4 --cpu-seconds 0 --no-roofline --no-legs "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', d['value'], d['ms_per_image'])" | tee -a gpurun_out/r5r/ab_224.txt
}
for rep in 1 2; do
  run tile224_off X=1 --
  run tile224_all OPP_TILE_224=2 --
  run tile224_big OPP_TILE_224=1 --
done
run s1_off X=1 -- --streams 1
run s1_all OPP_TILE_224=2 -- --streams 1
bash tools/pmc_conv.sh "layer2 3x3 196" gpurun_out/r5r/pmc128 --cfgs 22,27 > /dev/null 2>&1
python tools/pmc_conv_summary.py gpurun_out/r5r/pmc128 > gpurun_out/r5r/pmc128.txt
find gpurun_out/r5r/pmc128 -name "*.csv" -delete
grep -A30 "128, 224" gpurun_out/r5r/pmc128.txt | grep "SQ_WAVE_CYCLES\|SQ_INSTS_SALU\|ACTIVE_INST_MISC\|SQ_WAIT_ANY\|WAIT_INST_ANY\|128, 2"
