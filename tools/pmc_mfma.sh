#!/bin/bash
# MFMA-pipe utilisation of every kernel of the forward: one --pmc pass (--kernel-trace only) over bench.py, one forward in
# flight, fine branch on the same stream.  usage: tools/pmc_mfma.sh <outdir> ; then tools/pmc_mfma_summary.py <outdir> <csv>
OUT="$1"; mkdir -p "$OUT"; OUT="$(cd "$OUT" && pwd)"
cd /tmp && export TMPDIR=/tmp
OPP_FPN_OVERLAP=0 timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/mfma" -o p -- \
   python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --images-per-step 1 --cpu-seconds 0 --no-roofline --no-legs --streams 1 > "$OUT/mfma.log" 2>&1
ls "$OUT/mfma" | head
