#!/bin/bash
# PMC counter passes over the conv micro-bench (run on the GPU box via gpurun).
# usage: tools/pmc_conv.sh <only-filter> <outdir> [extra conv_bench args, e.g. --h2 1 --cfgs 0,20]
ONLY="$1"; OUT="$2"; shift; shift; EXTRA="$@"; mkdir -p "$OUT"; OUT="$(cd "$OUT" && pwd)"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$OUT/counters_list.txt" 2>&1
run() { # name, counters...
  name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -o p -- \
     python $GRAFT_REPO_ROOT/tools/conv_bench.py --only "$ONLY" --iters 3 $EXTRA > "$OUT/$name.log" 2>&1
}
run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
run p2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run p3 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_COUNT
run p4 FETCH_SIZE
run p5 WRITE_SIZE
run p6 TCC_HIT_sum TCC_MISS_sum
ls -R "$OUT" | head -40
