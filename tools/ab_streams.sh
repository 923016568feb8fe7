for rep in 1 2; do
for st in 2 3 4 5; do
 python bench.py --steps 40 --warmup 3 --cpu-seconds 0 --no-roofline --no-legs --streams $st 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams $st', d['value'], d['ms_per_image'])"
done
done
