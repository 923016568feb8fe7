#!/bin/bash
# K-loop parts of the 256x128 bf16x3 convolution tile removed several at a time (tuning library; timing only):
#   4xx = fp32 rows split in the K loop, 5xx = pre-split rows; xx = mask: 1 no global prefetch, 2 no LDS hand-over, 4 no barrier, 8 no fragment reads
#   (x15 = the bare MFMA sequence; x07 = + fragment reads; x03 = + barrier; x01 = + LDS hand-over; x00 = everything)
# bash tools/r06_abl2.sh > gpurun_out/r06_abl2.txt 2>&1
cd $GRAFT_REPO_ROOT
export OPP_HIP_LIB=$PWD/onepose_plus_plus_amd/libopp_hip_tuning.so OPP_ABLATE=1
for rep in 1 2; do
python tools/conv_bench.py --only "layer1 3x3" --cfgs 20,400,401,403,407,411,415 --iters 30
python tools/conv_bench.py --split 1 --only "layer1 3x3" --cfgs 20,500,501,503,504,507,508,511,515 --iters 30
done
