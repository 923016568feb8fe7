#!/bin/bash
# Pre-split activations, first GPU pass: kernel-level bit-identity tests, the backbone chain A/B, conv micro-bench split vs not.
# bash tools/r06_asp.sh > gpurun_out/r06_asp.log 2>&1
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py -q -x -k "presplit" 2>&1 | tail -5
python -m pytest tests/test_stages_gpu.py -q -x -k "presplit or backbone" 2>&1 | tail -5
for sp in 0 1 2 3; do
  echo "== split $sp"
  python tools/conv_bench.py --split $sp --iters 30 --only "3x3 128->128" --cfgs 20,25
  python tools/conv_bench.py --split $sp --iters 30 --only "l2_out2a" --cfgs 20,25
  python tools/conv_bench.py --split $sp --iters 30 --only "layer3 3x3" --cfgs 26,2
  python tools/conv_bench.py --split $sp --iters 30 --only "l1_outconv" --cfgs 22,25
  python tools/conv_bench.py --split $sp --iters 30 --only "layer2.0" --cfgs 22,25
done
