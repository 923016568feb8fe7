# headline workload, A/B of one environment switch inside one box (alternating repetitions):
#   bash tools/ab_env.sh OPP_QKV_SS 0 1 [-- extra bench.py args] > gpurun_out/ab.txt
VAR=$1; shift
VALS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do VALS+=("$1"); shift; done
[ "$1" = "--" ] && shift
for rep in 1 2 3; do
  for v in "${VALS[@]}"; do
    env $VAR=$v python bench.py --steps 40 --warmup 3 --cpu-seconds 0 --no-roofline --no-legs "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', d['value'], d['ms_per_image'])"
  done
done
