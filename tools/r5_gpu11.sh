set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5k
(timeout 900 python -m pytest tests/test_stages_gpu.py tests/test_kernels_gpu.py tests/test_train_bwd_gpu.py tests/test_assignmatrix.py -q --tb=short -p no:cacheprovider -x 2>&1 | tail -6) > gpurun_out/r5k/tests1.txt
(timeout 900 python -m pytest tests/test_e2e_gpu.py -q --tb=short -p no:cacheprovider -x -k "golden or full_size or two_sweep or ties" 2>&1 | tail -6) > gpurun_out/r5k/tests2.txt
python tools/matcher_bench.py --reps 60 > gpurun_out/r5k/matcher.txt 2>&1
python bench.py --steps 20 --warmup 5 --no-legs --cpu-seconds 0 --no-roofline > gpurun_out/r5k/bench.json 2> gpurun_out/r5k/bench.err
cat gpurun_out/r5k/tests1.txt gpurun_out/r5k/tests2.txt gpurun_out/r5k/matcher.txt
python -c "
import json; d=json.load(open('gpurun_out/r5k/bench.json')); print(d['value'], d['config']['streams_per_gpu'], d['config']['tile_policy'])"
