#!/bin/bash
# PMC counter passes over conv_wgrad_kernel at one training-step shape (run on the GPU box via gpurun).
# usage: tools/pmc_wgrad.sh <outdir> [shape substring, default "layer1 3x3"]
OUT="$1"; ONLY="${2:-layer1 3x3}"; mkdir -p "$OUT"; OUT="$(cd "$OUT" && pwd)"
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
  name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -o p -- \
     python $GRAFT_REPO_ROOT/tools/conv_bwd_bench.py 4 --only "$ONLY" --what wgrad > "$OUT/$name.log" 2>&1
}
run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
run p2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT
run p3 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES
run p4 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum
run p5 TA_BUSY_avr TA_BUFFER_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum
run p6 FETCH_SIZE
run p7 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
cd $GRAFT_REPO_ROOT
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(out + "/p*/")):
    files = glob.glob(d + "**/*counter_collection.csv", recursive=True)
    if not files:
        print(d, "no counter file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(files[0])):
        k = r["Kernel_Name"]
        if "conv_wgrad_kernel" not in k: continue
        acc[r["Counter_Name"]][r["Dispatch_Id"]].append(float(r["Counter_Value"]))
    for name, disp in acc.items():
        vals = [sum(v) for v in disp.values()]
        print("%-40s mean per dispatch %.4g  (%d dispatches)" % (name, sum(vals) / len(vals), len(vals)))
PY
rm -rf $OUT/p*/*trace.csv $OUT/p*/*/*trace.csv
