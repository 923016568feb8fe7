set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5s
(timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -25) > gpurun_out/r5s/gpu_tests.txt
cat gpurun_out/r5s/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/r5s/smoke.txt
