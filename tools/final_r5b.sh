# Round-5 closing evidence after the matcher / ring-tile changes (one gpurun call):  bash tools/final_r5b.sh > gpurun_out/final_r5b.log 2>&1
# PMC traffic first (bench.py reads profiles/traffic_symbols_bf16x3.json), then the bench lines, the two kernel traces, the matcher bench, smoke.
set -x
cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_kernels_gpu.py -q --tb=short -p no:cacheprovider -k "linear_bf16x3 or tile_shape or conv_bf16x3" 2>&1 | tail -6) > gpurun_out/final_kernel_tests.txt
cat gpurun_out/final_kernel_tests.txt
bash tools/pmc_bench.sh gpurun_out/final_pmc > /dev/null 2>&1
python tools/pmc_traffic_summary.py gpurun_out/final_pmc gpurun_out/final_pmc_traffic.csv profiles/traffic_symbols_bf16x3.json > /dev/null
cp profiles/traffic_symbols_bf16x3.json gpurun_out/final_traffic_symbols_bf16x3.json
python bench.py > gpurun_out/final_bench_default.json 2> gpurun_out/final_bench_default.err
python bench.py --steps 20 --warmup 5 > gpurun_out/final_bench_driver.json 2> gpurun_out/final_bench_driver.err
python tools/matcher_bench.py --reps 60 > gpurun_out/final_matcher_bench.txt 2>&1
cd /tmp && export TMPDIR=/tmp
OPP_FPN_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final_s1 -o s1 -- python $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 5 --images-per-step 4 --cpu-seconds 0 --no-legs --no-roofline --streams 1 > $GRAFT_REPO_ROOT/gpurun_out/final_s1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final_s3 -o s3 -- python $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 6 --images-per-step 4 --cpu-seconds 0 --no-legs --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/final_s3.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/final_s1 gpurun_out/final_s3 gpurun_out/final_pmc -name "*trace.csv" -delete
# in-box A/B of the 128 x 224 ring tile under the latency policy (one forward in flight), alternating
for rep in 1 2; do for v in 1 0; do
  OPP_TILE_224=$v python bench.py --steps 30 --warmup 4 --cpu-seconds 0 --no-roofline --no-legs --streams 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('one forward in flight, OPP_TILE_224=$v', d['value'], d['ms_per_image'])" >> gpurun_out/final_ab_tile224_latency.txt
done; done
cat gpurun_out/final_ab_tile224_latency.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.txt 2>&1
tail -2 gpurun_out/final_smoke.txt; cat gpurun_out/final_matcher_bench.txt | tail -3
python -c "
import json
for n in ('default','driver'):
    d=json.load(open('gpurun_out/final_bench_%s.json'%n)); r=d['roofline']
    print(n, d['value'], d['config']['model_frac_of_mfma_peak'], r['symbol'], r['us_per_forward'], r['frac'], r.get('traffic'))
"
