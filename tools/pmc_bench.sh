#!/bin/bash
# HBM-traffic PMC passes (separate runs, --kernel-trace only, as MI355X_MICROARCH.md prescribes) over
# the SAME bench command; run on the GPU box via gpurun.  usage: tools/pmc_bench.sh <outdir> [extra bench.py args]
OUT="$1"; shift; EXTRA="$@"; mkdir -p "$OUT"; OUT="$(cd "$OUT" && pwd)"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  OPP_FPN_OVERLAP=0 timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/$c" -o p -- \
     python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --images-per-step 1 --cpu-seconds 0 --no-roofline --no-legs --streams 1 $EXTRA > "$OUT/$c.log" 2>&1
done
ls -R "$OUT" | head
