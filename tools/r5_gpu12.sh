set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5l
(timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_stages_gpu.py -q --tb=short -p no:cacheprovider -x -k "encoder or transformer or chain or attention" 2>&1 | tail -8) > gpurun_out/r5l/tests1.txt
(timeout 900 python -m pytest tests/test_e2e_gpu.py -q --tb=short -p no:cacheprovider -x -k "(golden and bf16x3) or prefix or token_cache or masked or full_size or tile_policy or oracle_and_determinism" 2>&1 | tail -8) > gpurun_out/r5l/tests2.txt
python bench.py --steps 20 --warmup 5 --no-legs --cpu-seconds 0 > gpurun_out/r5l/bench_fold_on.json 2> gpurun_out/r5l/bench_fold_on.err
OPP_QKV_FOLD=0 python bench.py --steps 20 --warmup 5 --no-legs --cpu-seconds 0 --no-roofline > gpurun_out/r5l/bench_fold_off.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-legs --cpu-seconds 0 --no-roofline --streams 1 > gpurun_out/r5l/bench_fold_on_s1.json 2>/dev/null
OPP_QKV_FOLD=0 python bench.py --steps 20 --warmup 5 --no-legs --cpu-seconds 0 --no-roofline --streams 1 > gpurun_out/r5l/bench_fold_off_s1.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-legs --cpu-seconds 0 --no-roofline > gpurun_out/r5l/bench_fold_on2.json 2>/dev/null
OPP_QKV_FOLD=0 python bench.py --steps 20 --warmup 5 --no-legs --cpu-seconds 0 --no-roofline > gpurun_out/r5l/bench_fold_off2.json 2>/dev/null
cat gpurun_out/r5l/tests1.txt gpurun_out/r5l/tests2.txt
python - <<'PY'
import json
for n in ("bench_fold_on","bench_fold_off","bench_fold_on2","bench_fold_off2","bench_fold_on_s1","bench_fold_off_s1"):
    try:
        d=json.load(open('gpurun_out/r5l/%s.json'%n)); print(n, d['value'], d['config']['tile_policy'], d['config']['streams_per_gpu'])
    except Exception as e: print(n,'ERR',e)
d=json.load(open('gpurun_out/r5l/bench_fold_on.json'))
r=d['roofline']
print(r['kernel'][:70], r['us_per_forward'], r['frac'])
for k in r['other_kernels']: print(k['symbol'][:50], k['launches_per_forward'], k['avg_launch_us'], k['us_per_forward'], k['frac'])
PY
