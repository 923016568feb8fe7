# Re-take the bench lines, the kernel traces, the training probe and the GPU test tail after a code change that does not alter
# memory traffic (the PMC passes of tools/final_profiles.sh stay valid):  bash tools/final_refresh.sh > gpurun_out/final_refresh.log 2>&1
set -x
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/final_bench_default.json 2> gpurun_out/final_bench_default.err
python bench.py --steps 20 --warmup 5 > gpurun_out/final_bench_driver.json 2> gpurun_out/final_bench_driver.err
python tools/train_probe.py > gpurun_out/final_train_probe.txt 2> gpurun_out/final_train_probe.err
cd /tmp && export TMPDIR=/tmp
OPP_FPN_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final_s1 -o s1 -- python $GRAFT_REPO_ROOT/bench.py --steps 25 --warmup 5 --images-per-step 1 --cpu-seconds 0 --no-legs --streams 1 > $GRAFT_REPO_ROOT/gpurun_out/final_s1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final_s3 -o s3 -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 6 --images-per-step 1 --cpu-seconds 0 --no-legs > $GRAFT_REPO_ROOT/gpurun_out/final_s3.log 2>&1
cd $GRAFT_REPO_ROOT; rm -f gpurun_out/final_s1/*trace.csv gpurun_out/final_s3/*trace.csv
(timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -15) > gpurun_out/final_gpu_tests.txt
