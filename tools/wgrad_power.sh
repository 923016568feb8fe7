# is conv_wgrad_kernel power-limited?  gfx clock / socket power while it runs back to back (real operands), and the same kernel
# with cache-resident operands (tuning library, OPP_WGRAD_ABLATE=7).   bash tools/wgrad_power.sh > gpurun_out/wgrad_power.txt
export OPP_HIP_LIB=$PWD/onepose_plus_plus_amd/libopp_hip_tuning.so
for a in 0 7; do
  echo "== OPP_WGRAD_ABLATE=$a"
  OPP_WGRAD_ABLATE=$a python tools/smi_trace.py --out gpurun_out/wgrad_power_$a -- python tools/conv_bwd_bench.py 4 --only "layer1 3x3" --what wgrad --iters 6000 2>/dev/null | grep "layer1"
  python - <<PY
import json
s = json.load(open("gpurun_out/wgrad_power_$a.summary.json"))
print(json.dumps(s.get("amd_smi", s), indent=0)[:1500])
PY
done
echo "== dgrad (forward kernel) for comparison"
python tools/smi_trace.py --out gpurun_out/wgrad_power_dgrad -- python tools/conv_bwd_bench.py 4 --only "layer1 3x3" --what dgrad --iters 6000 2>/dev/null | grep layer1
python - <<PY
import json
s = json.load(open("gpurun_out/wgrad_power_dgrad.summary.json"))
print(json.dumps(s.get("amd_smi", s), indent=0)[:1500])
PY
