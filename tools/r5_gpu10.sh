set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5j
bash tools/ss_dephase_ab.sh > gpurun_out/r5j/ss_dephase.txt 2>&1
(timeout 600 python -m pytest tests/test_train_bwd_gpu.py tests/test_stages_gpu.py -q --tb=short -p no:cacheprovider -k "fine_head or matcher" 2>&1 | tail -6) > gpurun_out/r5j/tests.txt
cat gpurun_out/r5j/ss_dephase.txt gpurun_out/r5j/tests.txt
