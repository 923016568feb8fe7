# End-of-round evidence, one gpurun call:  bash tools/final_profiles.sh > gpurun_out/final_profiles.log 2>&1
# Order matters: the PMC passes come first, their summary is what bench.py's roofline leg reads as `traffic`.
set -x
cd $GRAFT_REPO_ROOT
bash tools/pmc_bench.sh gpurun_out/final_pmc
python tools/pmc_traffic_summary.py gpurun_out/final_pmc gpurun_out/final_pmc_traffic.csv profiles/traffic_symbols_bf16x3.json > /dev/null
cp profiles/traffic_symbols_bf16x3.json gpurun_out/final_traffic_symbols_bf16x3.json     # profiles/ does not travel back; gpurun_out/ does
bash tools/pmc_mfma.sh gpurun_out/final_pmc_mfma > /dev/null 2>&1
python tools/pmc_mfma_summary.py gpurun_out/final_pmc_mfma gpurun_out/final_pmc_mfma_util.csv > gpurun_out/final_pmc_mfma_util.txt 2>&1
python bench.py > gpurun_out/final_bench_default.json 2> gpurun_out/final_bench_default.err
python bench.py --steps 20 --warmup 5 > gpurun_out/final_bench_driver.json 2> gpurun_out/final_bench_driver.err
cp bench_detail.json gpurun_out/final_bench_detail_driver.json      # the sidecar of THAT line (every later bench run overwrites bench_detail.json)
python tools/smi_trace.py --out gpurun_out/final_smi -- python bench.py --steps 200 --warmup 5 --cpu-seconds 0 --no-legs --no-roofline > gpurun_out/final_smi.log 2>&1
python tools/train_probe.py > gpurun_out/final_train_probe.txt 2> gpurun_out/final_train_probe.err
python tools/conv_bwd_bench.py 4 > gpurun_out/final_conv_bwd_bench.txt 2>/dev/null
python tools/fine_leg.py > gpurun_out/final_fine_leg.json 2> gpurun_out/final_fine_leg.err
python tools/matcher_bench.py --reps 60 > gpurun_out/final_matcher_bench.txt 2>&1
cd /tmp && export TMPDIR=/tmp
# single stream, fine branch on the same stream, the tiles of the timed region: the condition of bench.py's roofline pass
OPP_FPN_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final_s1 -o s1 -- python $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 5 --images-per-step 4 --cpu-seconds 0 --no-legs --no-roofline --streams 1 > $GRAFT_REPO_ROOT/gpurun_out/final_s1.log 2>&1
# the timed region itself (default streams, throughput tiles; no roofline pass: its 500+ single-stream forwards would dominate the table)
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final_s3 -o s3 -- python $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 6 --images-per-step 4 --cpu-seconds 0 --no-legs --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/final_s3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final_fine -o fine -- python $GRAFT_REPO_ROOT/tools/fine_profile.py > $GRAFT_REPO_ROOT/gpurun_out/final_fine.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final_train -o tr -- python $GRAFT_REPO_ROOT/tools/train_trace.py 4 > $GRAFT_REPO_ROOT/gpurun_out/final_train.log 2>&1
cd $GRAFT_REPO_ROOT; rm -f gpurun_out/final_s1/*trace.csv gpurun_out/final_s3/*trace.csv gpurun_out/final_fine/*trace.csv gpurun_out/final_train/*trace.csv
find gpurun_out/final_s1 gpurun_out/final_s3 gpurun_out/final_fine gpurun_out/final_train -name "*trace.csv" -delete
rm -rf gpurun_out/final_pmc/FETCH_SIZE/*trace.csv gpurun_out/final_pmc/WRITE_SIZE/*trace.csv gpurun_out/final_pmc_mfma/mfma/*trace.csv
# (the GPU suite runs separately: tools/gpu_tests.sh)

ls gpurun_out/final_s1 gpurun_out/final_s3 gpurun_out/final_fine gpurun_out/final_pmc
