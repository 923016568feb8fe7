set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5a
(timeout 900 python -m pytest tests/test_e2e_gpu.py -q --tb=short -p no:cacheprovider -k "prefix or token_cache or tile_policy or matcher_pool or skipping or full_size" 2>&1 | tail -15) > gpurun_out/r5a/tests1.txt
(timeout 600 python -m pytest tests/test_train_bwd_gpu.py -q --tb=short -p no:cacheprovider -k "frozen or odd_size" 2>&1 | tail -15) > gpurun_out/r5a/tests2.txt
python bench.py --steps 20 --warmup 5 --no-legs --cpu-seconds 0 > gpurun_out/r5a/bench_driver.json 2> gpurun_out/r5a/bench_driver.err
OPP_HIP_LIB=$GRAFT_REPO_ROOT/onepose_plus_plus_amd/libopp_hip_tuning.so OPP_ABLATE=1 python tools/conv_bench.py --only 192 --iters 20 > gpurun_out/r5a/conv192.txt 2>&1
OPP_HIP_LIB=$GRAFT_REPO_ROOT/onepose_plus_plus_amd/libopp_hip_tuning.so OPP_ABLATE=1 python tools/conv_bench.py --only "layer2 3x3" --iters 20 >> gpurun_out/r5a/conv192.txt 2>&1
OPP_HIP_LIB=$GRAFT_REPO_ROOT/onepose_plus_plus_amd/libopp_hip_tuning.so OPP_ABLATE=1 python tools/conv_bench.py --only "l1_out2a 3x3" --iters 20 >> gpurun_out/r5a/conv192.txt 2>&1
cat gpurun_out/r5a/tests1.txt gpurun_out/r5a/tests2.txt gpurun_out/r5a/conv192.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5a/bench_driver.json'))
print(d['value'], d['ms_per_step'], d['config']['model_frac_of_mfma_peak'])
r=d['roofline']
print(r['kernel'][:60], r['us_per_forward'], r['frac'])
for k in r['other_kernels']: print(k['symbol'][:50], k['launches_per_forward'], k['avg_launch_us'], k['us_per_forward'], k['frac'])
PY
