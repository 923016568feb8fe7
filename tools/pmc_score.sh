#!/bin/bash
# HBM fetch / write of the matcher kernels alone (tools/matcher_bench.py), separate --pmc passes.  usage: tools/pmc_score.sh <outdir>
OUT="$1"; mkdir -p "$OUT"; OUT="$(cd "$OUT" && pwd)"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/$c" -o p -- python $GRAFT_REPO_ROOT/tools/matcher_bench.py --reps 5 > "$OUT/$c.log" 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- python $GRAFT_REPO_ROOT/tools/matcher_bench.py --reps 30 > "$OUT/stats.log" 2>&1
rm -f "$OUT"/stats/*trace.csv
