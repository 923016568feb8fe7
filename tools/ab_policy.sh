# headline workload, tile policy A/B inside one box (alternating repetitions):  bash tools/ab_policy.sh > gpurun_out/ab_policy.txt
for rep in 1 2 3; do
  for pol in latency throughput; do
    python bench.py --steps 40 --warmup 3 --cpu-seconds 0 --no-roofline --no-legs --tile-policy $pol 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$pol', d['value'], d['ms_per_image'])"
  done
done
