set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5o
(timeout 600 python -m pytest tests/test_stages_gpu.py -q --tb=short -p no:cacheprovider -x -k "matcher or two_sweep or tiny" 2>&1 | tail -5) > gpurun_out/r5o/tests1.txt
(timeout 900 python -m pytest tests/test_e2e_gpu.py -q --tb=short -p no:cacheprovider -x -k "(golden and bf16x3) or 15000 or masked or oracle_and_determinism or batch" 2>&1 | tail -5) > gpurun_out/r5o/tests2.txt
cat gpurun_out/r5o/tests1.txt gpurun_out/r5o/tests2.txt
python tools/matcher_bench.py --reps 50 2>/dev/null | tee gpurun_out/r5o/matcher_bench.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5o/tr -o m -- python $GRAFT_REPO_ROOT/tools/matcher_bench.py --reps 50 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r5o/tr/**/*kernel_stats.csv', recursive=True)
rows=list(csv.DictReader(open(f[0])))
for r in rows[:16]: print("  %-70s calls %5s avg %8.2f us" % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
