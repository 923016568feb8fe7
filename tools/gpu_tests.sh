#!/bin/bash
# The whole GPU suite + smoke in one gpurun call (what the driver runs at round end):  bash tools/gpu_tests.sh
cd "$GRAFT_REPO_ROOT"
(timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -12) > gpurun_out/final_gpu_tests.txt
cat gpurun_out/final_gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/final_smoke.txt
