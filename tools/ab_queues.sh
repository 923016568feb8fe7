# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); two streams on one queue serialise.  A/B of the queue count
# and of the fine-branch side streams on the headline workload (3 forwards in flight):  bash tools/ab_queues.sh [extra bench args]
for rep in 1 2; do
  for q in 4 8; do
    for ov in 1 0; do
      GPU_MAX_HW_QUEUES=$q OPP_FPN_OVERLAP=$ov python bench.py --steps 40 --warmup 3 --cpu-seconds 0 --no-roofline --no-legs "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('queues $q overlap $ov', d['value'], d['ms_per_image'])"
    done
  done
done
