for f in 1 0; do
for st in 1 3; do
echo "== fpn_overlap $f streams $st"; OPP_FPN_OVERLAP=$f python bench.py --steps 8 --streams $st --no-roofline --no-legs --cpu-seconds 0 2>&1 | python -c "import sys,json; [print(json.loads(l)['value'], json.loads(l)['ms_per_image']) for l in sys.stdin if l.startswith('{')]"
done; done
echo "== full coarse-to-fine, single stream"
for f in 1 0; do OPP_FPN_OVERLAP=$f python bench.py --steps 4 --streams 1 --fine --thr 0.0 --no-roofline --no-legs --cpu-seconds 0 2>&1 | python -c "import sys,json; [print(json.loads(l)['value'], json.loads(l)['ms_per_image']) for l in sys.stdin if l.startswith('{')]"; done
