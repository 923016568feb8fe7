#!/bin/bash
# PMC counter passes over the gemm_ss probe (tuning build). usage: tools/pmc_ss.sh <outdir>
OUT="$1"; mkdir -p "$OUT"; OUT="$(cd "$OUT" && pwd)"
export OPP_HIP_LIB=$GRAFT_REPO_ROOT/onepose_plus_plus_amd/libopp_hip_tuning.so
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -o p -- python $GRAFT_REPO_ROOT/tools/gemm_ss_probe.py > "$OUT/$name.log" 2>&1
}
run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
run p2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run p3 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_COUNT
