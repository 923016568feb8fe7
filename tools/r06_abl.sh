#!/bin/bash
# Loop ablations of the bf16x3 convolution tiles on the tuning library (timing only; 39x configs = 256x128, 29x = 128x128):
#   120/122 phase-stamped baseline, x91 no global loads, x92 no LDS stores / split, x93 no barrier, x94 no fragment reads, x95 no split arithmetic
# bash tools/r06_abl.sh > gpurun_out/r06_abl.txt 2>&1
cd $GRAFT_REPO_ROOT
export OPP_HIP_LIB=$PWD/onepose_plus_plus_amd/libopp_hip_tuning.so OPP_ABLATE=1
python tools/conv_bench.py --only "layer1 3x3" --cfgs 20,120,391,392,393,394,395 --iters 30
python tools/conv_bench.py --only "l2_out2a" --cfgs 25,122,291,292,293,294,295 --iters 30
python tools/conv_bench.py --only "layer1 3x3" --cfgs 20,120,391,392,393,394,395 --iters 30
