set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
(timeout 900 python -m pytest tests/test_e2e_gpu.py -q --tb=short -p no:cacheprovider -k "match_driven or high_confidence or prefix" 2>&1 | tail -25) > gpurun_out/r5b/tests1.txt
(timeout 600 python -m pytest tests/test_train_bwd_gpu.py -q --tb=short -p no:cacheprovider -k "frozen" 2>&1 | tail -15) > gpurun_out/r5b/tests2.txt
python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > gpurun_out/r5b/bench_legs.json 2> gpurun_out/r5b/bench_legs.err
OPP_HIP_LIB=$GRAFT_REPO_ROOT/onepose_plus_plus_amd/libopp_hip_tuning.so OPP_ABLATE=1 python tools/conv_bench.py --only 192 --iters 30 > gpurun_out/r5b/conv192.txt 2>&1
OPP_HIP_LIB=$GRAFT_REPO_ROOT/onepose_plus_plus_amd/libopp_hip_tuning.so OPP_ABLATE=1 python tools/conv_bench.py --only 192 --iters 30 >> gpurun_out/r5b/conv192.txt 2>&1
cat gpurun_out/r5b/tests1.txt gpurun_out/r5b/tests2.txt gpurun_out/r5b/conv192.txt
tail -5 gpurun_out/r5b/bench_legs.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5b/bench_legs.json'))
print(d['value'], d['ms_per_step'], d['config']['model_frac_of_mfma_peak'])
print(json.dumps(d['roofline'].get('legs'), indent=1)[:6000])
PY
