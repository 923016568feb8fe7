# kernel-trace time of the l1_out lateral (1x1 128->196 @256^2 + bilinear residual) under different tiles
cd /tmp && export TMPDIR=/tmp
for cfg in -1 26 25 20 2; do
  rm -rf /tmp/l1o; OPP_FPN_OVERLAP=0 OPP_L1OUT_CFG=$cfg rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/l1o -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --images-per-step 1 --cpu-seconds 0 --no-legs --no-roofline --streams 1 > /tmp/l1o.log 2>&1
  echo "== OPP_L1OUT_CFG=$cfg  $(tail -1 /tmp/l1o.log | python -c 'import json,sys; print(json.loads(sys.stdin.read())["value"])')"
  python - <<'PY'
import csv, glob
f = glob.glob('/tmp/l1o/**/*kernel_stats.csv', recursive=True)
tot = 0
for row in csv.DictReader(open(f[0])):
    tot += float(row['TotalDurationNs'])
    n = row['Name']
    if 'opp_gemm_kernel' in n and 'true' in n:
        print("   %-70s calls %5s avg %8.2f us total %9.1f us" % (n[30:100], row['Calls'], float(row['AverageNs']) / 1e3, float(row['TotalDurationNs']) / 1e3))
print("   all kernels: %.1f us" % (tot / 1e3))
PY
done
