#!/bin/bash
# Coarse matcher alone: tests, tools/matcher_bench.py (whole opp_coarse_match between events) and its kernel trace (per-kernel averages).
#     bash tools/matcher_trace.sh > gpurun_out/matcher_trace.log 2>&1    (numbers behind profiles/r05_matcher_bench.txt)
set -x
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/matcher_trace; mkdir -p $OUT
(timeout 600 python -m pytest tests/test_stages_gpu.py -q --tb=short -p no:cacheprovider -x -k "matcher or two_sweep or tiny" 2>&1 | tail -5) | tee $OUT/tests.txt
python tools/matcher_bench.py --reps 50 2>/dev/null | tee $OUT/matcher_bench.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/tr" -o m -- python "$GRAFT_REPO_ROOT/tools/matcher_bench.py" --reps 50 > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"
find $OUT/tr -name "*trace.csv" -delete
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/matcher_trace/tr/**/*kernel_stats.csv', recursive=True)
for r in list(csv.DictReader(open(f[0])))[:16]:
    print("  %-78s calls %5s avg %8.2f us" % (r['Name'][:78], r['Calls'], float(r['AverageNs']) / 1e3))
PY
