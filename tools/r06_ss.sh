#!/bin/bash
# persistent score GEMM: parity, A/B against the one-tile kernel, timeline.   bash tools/r06_ss.sh > gpurun_out/r06_ss.log 2>&1
set -x
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_stages_gpu.py -q --tb=short -p no:cacheprovider -k "matcher or two_sweep or coarse" 2>&1 | tail -8) > gpurun_out/r06_ss_tests.txt
cat gpurun_out/r06_ss_tests.txt
for rep in 1 2; do
for v in "OPP_SS_PERSIST=0" "OPP_SS_PERSIST=1" "OPP_SS_PERSIST=1 OPP_SS_DELAY=0" "OPP_SS_PERSIST=1 OPP_SS_DELAY=6144" "OPP_SS_PERSIST=1 OPP_SS_DELAY=20480"; do
  echo "== $v" >> gpurun_out/r06_ss_ab.txt
  env $v python tools/matcher_bench.py --reps 60 2>&1 | grep "two_sweep=2" >> gpurun_out/r06_ss_ab.txt
done; done
cat gpurun_out/r06_ss_ab.txt
export OPP_HIP_LIB=$GRAFT_REPO_ROOT/onepose_plus_plus_amd/libopp_hip_tuning.so
for v in "OPP_SS_PERSIST=0" "OPP_SS_PERSIST=1" "OPP_SS_PERSIST=1 OPP_SS_DELAY=0"; do
  echo "== $v" >> gpurun_out/r06_ss_timeline.txt
  env $v python tools/ss_timeline.py >> gpurun_out/r06_ss_timeline.txt 2>&1
done
cat gpurun_out/r06_ss_timeline.txt
