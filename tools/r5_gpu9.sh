set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5i
(timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -25) > gpurun_out/r5i/gpu_tests_full.txt
for S in 2 3 4 5; do
  python bench.py --steps 20 --warmup 5 --no-legs --cpu-seconds 0 --no-roofline --streams $S > gpurun_out/r5i/bench_s$S.json 2> gpurun_out/r5i/bench_s$S.err
done
cat gpurun_out/r5i/gpu_tests_full.txt
python - <<'PY'
import json
for s in (2,3,4,5):
    try:
        d=json.load(open('gpurun_out/r5i/bench_s%d.json'%s)); print('streams',s, d['value'], d['config']['tile_policy'])
    except Exception as e: print(s,'ERR',e)
PY
