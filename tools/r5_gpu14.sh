set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5n
(timeout 600 python -m pytest tests/test_stages_gpu.py -q --tb=short -p no:cacheprovider -x -k "matcher or two_sweep or tiny" 2>&1 | tail -5) > gpurun_out/r5n/tests1.txt
(OPP_MATCH_FUSED=1 timeout 600 python -m pytest tests/test_stages_gpu.py -q --tb=short -p no:cacheprovider -x -k "matcher or two_sweep or tiny" 2>&1 | tail -5) > gpurun_out/r5n/tests1_fused.txt
cat gpurun_out/r5n/tests1.txt gpurun_out/r5n/tests1_fused.txt
for rep in 1 2; do
python tools/matcher_bench.py --reps 50 2>/dev/null | head -1 | tee -a gpurun_out/r5n/matcher_bench.txt
OPP_MATCH_FUSED=1 python tools/matcher_bench.py --reps 50 2>/dev/null | head -1 | sed 's/^/fused /' | tee -a gpurun_out/r5n/matcher_bench.txt
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5n/tr -o m -- python $GRAFT_REPO_ROOT/tools/matcher_bench.py --reps 50 > /dev/null 2>&1
OPP_MATCH_FUSED=1 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5n/trf -o m -- python $GRAFT_REPO_ROOT/tools/matcher_bench.py --reps 50 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob
for d in ("tr","trf"):
    f=glob.glob('gpurun_out/r5n/%s/**/*kernel_stats.csv'%d, recursive=True)
    if not f: print(d,'no stats'); continue
    rows=list(csv.DictReader(open(f[0])))
    print(d)
    for r in rows[:16]: print("  %-70s calls %5s avg %8.2f us" % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
