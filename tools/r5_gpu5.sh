set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5g
(timeout 600 python -m pytest tests/test_stages_gpu.py tests/test_e2e_gpu.py tests/test_train_bwd_gpu.py -q --tb=short -p no:cacheprovider -x -k "backbone or (golden and bf16x3) or match_driven or conv" 2>&1 | tail -8) > gpurun_out/r5g/tests1.txt
python bench.py --steps 20 --warmup 5 --no-legs --cpu-seconds 0 > gpurun_out/r5g/bench_tail_on.json 2> gpurun_out/r5g/bench_tail_on.err
OPP_CONV_TAIL=0 python bench.py --steps 20 --warmup 5 --no-legs --cpu-seconds 0 --no-roofline > gpurun_out/r5g/bench_tail_off.json 2> gpurun_out/r5g/bench_tail_off.err
python bench.py --steps 20 --warmup 5 --no-legs --cpu-seconds 0 --no-roofline --streams 1 > gpurun_out/r5g/bench_tail_on_s1.json 2>/dev/null
OPP_CONV_TAIL=0 python bench.py --steps 20 --warmup 5 --no-legs --cpu-seconds 0 --no-roofline --streams 1 > gpurun_out/r5g/bench_tail_off_s1.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5g/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --images-per-step 1 --cpu-seconds 0 --no-legs --no-roofline --streams 1 > $GRAFT_REPO_ROOT/gpurun_out/r5g/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/tail_trace.py gpurun_out/r5g/trace > gpurun_out/r5g/tail_trace.txt 2>&1
rm -rf gpurun_out/r5g/trace
cat gpurun_out/r5g/tests1.txt gpurun_out/r5g/tail_trace.txt
python - <<'PY'
import json
for n in ("bench_tail_on","bench_tail_off","bench_tail_on_s1","bench_tail_off_s1"):
    try:
        d=json.load(open('gpurun_out/r5g/%s.json'%n))
        print(n, d['value'], d['config']['tile_policy'], d['config']['model_frac_of_mfma_peak'])
    except Exception as e: print(n, 'ERR', e)
d=json.load(open('gpurun_out/r5g/bench_tail_on.json'))
r=d['roofline']
print(r['kernel'][:70], r['us_per_forward'], r['frac'])
for k in r['other_kernels']: print(k['symbol'][:50], k['launches_per_forward'], k['avg_launch_us'], k['us_per_forward'], k['frac'])
PY
