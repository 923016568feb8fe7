# A/B of the SLP vectorizer in gemm_mfma.hip (v_pk_add_f32 beside the MFMAs of the bf16 split): images/s of the headline bench with the
# shipped library (no SLP in gemm_mfma.hip) against `python -m onepose_plus_plus_amd.build --variant slp`
for r in 1 2 3; do
  for v in default slp; do
    if [ $v = slp ]; then export OPP_HIP_LIB=$PWD/onepose_plus_plus_amd/libopp_hip_slp.so; else unset OPP_HIP_LIB; fi
    python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-legs --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])"
  done
done
