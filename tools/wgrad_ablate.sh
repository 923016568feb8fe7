# where conv_wgrad_kernel spends its time: the layer1 weight gradient (B = 4) with parts of the loop removed (timing only, results wrong)
for a in 0 1 2 3 4 0; do
  echo "== OPP_WGRAD_ABLATE=$a"; OPP_WGRAD_ABLATE=$a python tools/conv_bwd_bench.py 4 2>/dev/null | grep -E "layer1 3x3|l2_out2a|l1_out2a" | sed 's/dgrad.*wgrad/wgrad/'
done
