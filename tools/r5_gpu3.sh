set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5c
(timeout 900 python -m pytest tests/test_e2e_gpu.py tests/test_assignmatrix.py -q --tb=short -p no:cacheprovider -k "match_driven or assignmatrix or hip_builder or fine_branch_overlap or skipping" 2>&1 | tail -25) > gpurun_out/r5c/tests1.txt
python tools/fine_leg.py > gpurun_out/r5c/fine_leg.json 2> gpurun_out/r5c/fine_leg.err
OPP_HIP_LIB=$GRAFT_REPO_ROOT/onepose_plus_plus_amd/libopp_hip_tuning.so OPP_ABLATE=1 python tools/conv_bench.py --only 192 --iters 30 > gpurun_out/r5c/conv192.txt 2>&1
OPP_HIP_LIB=$GRAFT_REPO_ROOT/onepose_plus_plus_amd/libopp_hip_tuning.so OPP_ABLATE=1 python tools/conv_bench.py --only 192 --iters 30 >> gpurun_out/r5c/conv192.txt 2>&1
cat gpurun_out/r5c/tests1.txt gpurun_out/r5c/conv192.txt
tail -3 gpurun_out/r5c/fine_leg.err
cat gpurun_out/r5c/fine_leg.json
