# kernel-trace durations of the coarse matcher's kernels under the OPP_SS_PRIO modes:  bash tools/ss_prio_ab.sh > gpurun_out/ss_prio.txt
cd /tmp && export TMPDIR=/tmp
for mode in 0 1 2 0 1; do
  rm -rf /tmp/ssp; OPP_SS_PRIO=$mode rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ssp -o t -- python $GRAFT_REPO_ROOT/tools/matcher_bench.py --reps 40 > /tmp/ssp.log 2>&1
  echo "== OPP_SS_PRIO=$mode"; grep "two_sweep=2" /tmp/ssp.log
  python - <<'PY'
import csv, glob
f = glob.glob('/tmp/ssp/**/*kernel_stats.csv', recursive=True)
for row in csv.DictReader(open(f[0])):
    n = row['Name']
    if any(k in n for k in ('gemm_ss_kernel', 'conf_reg', 'select_kernel', 'col_max', 'row_merge', 'col_merge', 'b3_split')):
        print("   %-60s calls %5s avg %8.2f us" % (n[:60], row['Calls'], float(row['AverageNs']) / 1e3))
PY
done
