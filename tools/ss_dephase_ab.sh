# kernel-trace durations of the score GEMM under the OPP_SS_PRIO de-phasing modes (3 / 4 / 5 = 8 / 16 / 24 k cycles of initial sleep for the second
# resident workgroup of every CU):  bash tools/ss_dephase_ab.sh > gpurun_out/ss_dephase.txt
cd /tmp && export TMPDIR=/tmp
for mode in 0 3 4 5 0 4; do
  rm -rf /tmp/ssp; OPP_SS_PRIO=$mode rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ssp -o t -- python $GRAFT_REPO_ROOT/tools/matcher_bench.py --reps 60 > /tmp/ssp.log 2>&1
  echo "== OPP_SS_PRIO=$mode"; grep "two_sweep=2" /tmp/ssp.log
  python - <<'PY'
import csv, glob
f = glob.glob('/tmp/ssp/**/*kernel_stats.csv', recursive=True)
for row in csv.DictReader(open(f[0])):
    n = row['Name']
    if any(k in n for k in ('gemm_ss_kernel<3>', 'gemm_ss_kernel<(int)3>', 'conf_reg')):
        print("   %-60s calls %5s avg %8.2f us" % (n[:60], row['Calls'], float(row['AverageNs']) / 1e3))
PY
done
