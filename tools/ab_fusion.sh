for f in 1 0; do
echo "== fusion $f streams 3"; OPP_ENCODER_FUSION=$f python bench.py --steps 6 --no-roofline --no-legs --cpu-seconds 0 2>&1 | python -c "import sys,json; [print(json.loads(l)['value'], json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]"
echo "== fusion $f streams 1"; OPP_ENCODER_FUSION=$f python bench.py --steps 6 --streams 1 --no-roofline --no-legs --cpu-seconds 0 2>&1 | python -c "import sys,json; [print(json.loads(l)['value'], json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]"
done
