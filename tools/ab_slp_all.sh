# A/B: every source without the SLP vectorizer (`python -m onepose_plus_plus_amd.build --variant noslp_all`) against the shipped library
# (only gemm_mfma.hip / conv_bwd.hip without it): the kernel tests (bit-identity of the fused encoder chain among them), then images/s
export V=$PWD/onepose_plus_plus_amd/libopp_hip_noslp_all.so
OPP_HIP_LIB=$V python -m pytest tests/test_kernels_gpu.py -q 2>&1 | tail -3 | cut -c1-200
for r in 1 2; do
  for v in default noslp_all; do
    if [ $v = noslp_all ]; then export OPP_HIP_LIB=$V; else unset OPP_HIP_LIB; fi
    python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-legs --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])"
  done
done
