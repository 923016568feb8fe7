set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5h
(timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_stages_gpu.py tests/test_loss_gpu.py -q --tb=short -p no:cacheprovider -x 2>&1 | tail -8) > gpurun_out/r5h/tests1.txt
(timeout 900 python -m pytest tests/test_e2e_gpu.py -q --tb=short -p no:cacheprovider -k "conv_tail or prefix or (golden and bf16x3) or token_cache or match_driven_fine_branch_is_bit" 2>&1 | tail -8) > gpurun_out/r5h/tests2.txt
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 tools/train_scale.py --steps 3 --warmup 1 > gpurun_out/r5h/train_scale_n1.json 2> gpurun_out/r5h/train_scale_n1.err
python bench.py --gpus 1 --fine --thr 0 --n-points 15000 --steps 5 --warmup 2 --no-legs --no-roofline --cpu-seconds 0 > gpurun_out/r5h/cfg3_n1.json 2> gpurun_out/r5h/cfg3_n1.err
python bench.py --steps 20 --warmup 5 --no-legs --cpu-seconds 0 > gpurun_out/r5h/bench.json 2> gpurun_out/r5h/bench.err
cat gpurun_out/r5h/tests1.txt gpurun_out/r5h/tests2.txt
tail -3 gpurun_out/r5h/train_scale_n1.err; cat gpurun_out/r5h/train_scale_n1.json
tail -2 gpurun_out/r5h/cfg3_n1.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5h/cfg3_n1.json')); print('cfg3', d['value'], d['config']['matches_last_step'], d['config']['workload'][:80])
d=json.load(open('gpurun_out/r5h/bench.json'))
print(d['value'], d['config']['tile_policy'], d['config']['model_frac_of_mfma_peak'])
r=d['roofline']
print(r['kernel'][:70], r['us_per_forward'], r['frac'])
for k in r['other_kernels']: print(k['symbol'][:50], k['launches_per_forward'], k['avg_launch_us'], k['us_per_forward'], k['frac'])
PY
