set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5d
(timeout 900 python -m pytest tests/test_stages_gpu.py tests/test_kernels_gpu.py -q --tb=short -p no:cacheprovider -x 2>&1 | tail -15) > gpurun_out/r5d/tests1.txt
(timeout 900 python -m pytest tests/test_e2e_gpu.py tests/test_train_bwd_gpu.py -q --tb=short -p no:cacheprovider -k "(golden and bf16x3) or tile_policy or match_driven or training_step_gradients or backbone_node or full_size or train_mode" 2>&1 | tail -15) > gpurun_out/r5d/tests2.txt
python bench.py --steps 20 --warmup 5 --no-legs --cpu-seconds 0 > gpurun_out/r5d/bench_tail_on.json 2> gpurun_out/r5d/bench_tail_on.err
OPP_CONV_TAIL=0 python bench.py --steps 20 --warmup 5 --no-legs --cpu-seconds 0 --no-roofline > gpurun_out/r5d/bench_tail_off.json 2> gpurun_out/r5d/bench_tail_off.err
python bench.py --steps 20 --warmup 5 --no-legs --cpu-seconds 0 --no-roofline --tile-policy latency > gpurun_out/r5d/bench_tail_on_latency.json 2>/dev/null
OPP_CONV_TAIL=0 python bench.py --steps 20 --warmup 5 --no-legs --cpu-seconds 0 --no-roofline --tile-policy latency > gpurun_out/r5d/bench_tail_off_latency.json 2>/dev/null
cat gpurun_out/r5d/tests1.txt gpurun_out/r5d/tests2.txt
tail -3 gpurun_out/r5d/bench_tail_on.err
python - <<'PY'
import json
for n in ("bench_tail_on","bench_tail_off","bench_tail_on_latency","bench_tail_off_latency"):
    try:
        d=json.load(open('gpurun_out/r5d/%s.json'%n))
        print(n, d['value'], d['config']['tile_policy'], d['config']['model_frac_of_mfma_peak'])
    except Exception as e: print(n, 'ERR', e)
d=json.load(open('gpurun_out/r5d/bench_tail_on.json'))
r=d['roofline']
print(r['kernel'][:70], r['us_per_forward'], r['frac'], r.get('launch_shapes') and [ (x['launches_per_forward'],x['avg_launch_us'],x['frac']) for x in r['launch_shapes']])
for k in r['other_kernels']: print(k['symbol'][:50], k['launches_per_forward'], k['avg_launch_us'], k['us_per_forward'], k['frac'])
PY
