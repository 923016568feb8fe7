#!/bin/bash
# Training step at the current code state: torch.profiler table (tools/train_probe.py) and the rocprofv3 kernel table of whole steps (tools/train_trace.py)
#     bash tools/train_refresh.sh      ->  gpurun_out/final_train_probe.txt, gpurun_out/final_train/*kernel_stats.csv
cd "$GRAFT_REPO_ROOT"
python tools/train_probe.py > gpurun_out/final_train_probe.txt 2> gpurun_out/final_train_probe.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/final_train" -o tr -- python "$GRAFT_REPO_ROOT/tools/train_trace.py" 4 > "$GRAFT_REPO_ROOT/gpurun_out/final_train.log" 2>&1
cd "$GRAFT_REPO_ROOT"; find gpurun_out/final_train -name "*trace.csv" -delete
head -3 gpurun_out/final_train_probe.txt; tail -2 gpurun_out/final_train.log
