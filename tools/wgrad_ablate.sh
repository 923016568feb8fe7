# where conv_wgrad_kernel spends its time: weight gradients at B = 4 with one part of the chunk body compiled out
# (tuning library, timing only: the ablated results are wrong).  bash tools/wgrad_ablate.sh > gpurun_out/wgrad_ablate.txt
export OPP_HIP_LIB=$PWD/onepose_plus_plus_amd/libopp_hip_tuning.so
names=("full kernel" "no split arithmetic" "no global loads" "no MFMAs" "no LDS hand-over" "no fragment reads" "no barrier in the loop")
for a in 0 1 2 3 4 5 6 0; do
  echo "== OPP_WGRAD_ABLATE=$a (${names[$a]})"
  OPP_WGRAD_ABLATE=$a python tools/conv_bwd_bench.py 4 2>/dev/null | grep -E "layer1 3x3|l2_out2a|l1_out2a" | sed 's/dgrad.*wgrad/wgrad/'
done
