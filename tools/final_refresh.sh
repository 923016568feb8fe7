# Re-take the bench lines and the kernel traces after a change that does not alter memory traffic (the PMC passes of tools/final_profiles.sh stay
# valid):  bash tools/final_refresh.sh > gpurun_out/final_refresh.log 2>&1
set -x
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/final_bench_default.json 2> gpurun_out/final_bench_default.err
python bench.py --steps 20 --warmup 5 > gpurun_out/final_bench_driver.json 2> gpurun_out/final_bench_driver.err
cd /tmp && export TMPDIR=/tmp
OPP_FPN_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final_s1 -o s1 -- python $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 5 --images-per-step 4 --cpu-seconds 0 --no-legs --no-roofline --streams 1 > $GRAFT_REPO_ROOT/gpurun_out/final_s1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final_s3 -o s3 -- python $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 6 --images-per-step 4 --cpu-seconds 0 --no-legs --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/final_s3.log 2>&1
cd $GRAFT_REPO_ROOT; find gpurun_out/final_s1 gpurun_out/final_s3 -name "*trace.csv" -delete
