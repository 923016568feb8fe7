#!/bin/bash
# A/B: pre-split activation chain of the bf16x3 backbone (OPP_ASP=1, default) against the fp32-row chain (OPP_ASP=0); headline configuration and one stream.
# bash tools/ab_asp.sh > gpurun_out/r06_ab_asp.txt 2>&1
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for asp in 1 0; do
  echo "== OPP_ASP=$asp (4 streams)"
  OPP_ASP=$asp python bench.py --steps 20 --warmup 5 --no-legs --no-roofline --cpu-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_image'])"
  echo "== OPP_ASP=$asp (1 stream)"
  OPP_ASP=$asp python bench.py --steps 20 --warmup 5 --no-legs --no-roofline --cpu-seconds 0 --streams 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_image'])"
done
done
