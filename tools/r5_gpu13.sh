# r05: balanced conf_reg + fused column-max / selection (matcher), then scheduling A/B of the headline (streams, split-K, conv tail)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5m
(timeout 600 python -m pytest tests/test_stages_gpu.py -q --tb=short -p no:cacheprovider -x -k "matcher or two_sweep or tiny" 2>&1 | tail -12) > gpurun_out/r5m/tests1.txt
(timeout 900 python -m pytest tests/test_e2e_gpu.py -q --tb=short -p no:cacheprovider -x -k "(golden and bf16x3) or n15000 or 15000 or masked or oracle_and_determinism or batch" 2>&1 | tail -12) > gpurun_out/r5m/tests2.txt
cat gpurun_out/r5m/tests1.txt gpurun_out/r5m/tests2.txt
python tools/matcher_bench.py --reps 50 2>/dev/null | tee gpurun_out/r5m/matcher_bench.txt
run() { # label, env..., -- args
  local label=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --steps 30 --warmup 4 --cpu-seconds 0 --no-roofline --no-legs "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', d['value'], d['ms_per_image'])" | tee -a gpurun_out/r5m/ab_sched.txt
}
for rep in 1 2; do
  run base X=1 --
  run splitk_off OPP_CONV_SPLITK=0 --
  run streams5 X=1 -- --streams 5
  run streams6 X=1 -- --streams 6
  run streams6_q8 GPU_MAX_HW_QUEUES=8 -- --streams 6
  run conv_tail OPP_CONV_TAIL=1 --
  run splitk_off_tail OPP_CONV_SPLITK=0 OPP_CONV_TAIL=1 --
done
python bench.py --steps 20 --warmup 5 --no-legs --cpu-seconds 0 > gpurun_out/r5m/bench_roofline.json 2> gpurun_out/r5m/bench_roofline.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5m/bench_roofline.json'))
r=d['roofline']
print(d['value'], d['config']['model_frac_of_mfma_peak'])
print(r['kernel'][:70], r['us_per_forward'], r['frac'])
for k in r['other_kernels']: print(k['symbol'][:50], k['launches_per_forward'], k['avg_launch_us'], k['us_per_forward'], k['frac'])
PY
