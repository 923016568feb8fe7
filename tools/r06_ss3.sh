#!/bin/bash
# score GEMM with three resident workgroups per CU (gemm_ss_res3_kernel, OPP_SS_RES3): matcher tests, matcher A/B, MFMA-pipe counter, kernel trace, forward A/B.
# bash tools/r06_ss3.sh > gpurun_out/r06_ss3.log 2>&1
cd $GRAFT_REPO_ROOT
echo "== tests OPP_SS_RES3=1"
OPP_SS_RES3=1 timeout 900 python -m pytest tests/test_stages_gpu.py tests/test_e2e_gpu.py -q --tb=short -p no:cacheprovider -k "matcher or two_sweep or coarse or highconf or (e2e_vs_golden and bf16x3)" 2>&1 | tail -2
for rep in 1 2 3; do
for v in "OPP_SS_RES3=0" "OPP_SS_RES3=1"; do
  echo "== $v"
  env $v python tools/matcher_bench.py --reps 60 2>&1 | grep "two_sweep=2"
done; done
for v in 0 1; do
  OPP_SS_RES3=$v bash tools/pmc_mfma.sh gpurun_out/r06_ss3_pmc_$v > /dev/null 2>&1
  python tools/pmc_mfma_summary.py gpurun_out/r06_ss3_pmc_$v gpurun_out/r06_ss3_pmc_$v.csv > /dev/null 2>&1
  echo "== PMC OPP_SS_RES3=$v"; grep "gemm_ss" gpurun_out/r06_ss3_pmc_$v.csv
  rm -rf gpurun_out/r06_ss3_pmc_$v
done
for rep in 1 2; do
for v in 0 1; do
  echo "== forward OPP_SS_RES3=$v (4 streams / 1 stream)"
  OPP_SS_RES3=$v python bench.py --steps 20 --warmup 5 --no-legs --no-roofline --cpu-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_image'])"
  OPP_SS_RES3=$v python bench.py --steps 20 --warmup 5 --no-legs --no-roofline --cpu-seconds 0 --streams 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_image'])"
done; done
