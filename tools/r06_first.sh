#!/bin/bash
# Round-6 opening call: bench line (compact, driver style) + sidecar, conv LDS counters, full GPU suite.   bash tools/r06_first.sh > gpurun_out/r06_first.log 2>&1
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_driver.json 2> gpurun_out/r06_bench_driver.err
wc -c gpurun_out/r06_bench_driver.json; cat gpurun_out/r06_bench_driver.json
cp bench_detail.json gpurun_out/r06_bench_detail_driver.json
bash tools/pmc_conv.sh "layer1 3x3" gpurun_out/r06_pmc_conv_256x128 --cfgs 20 > /dev/null 2>&1
python tools/pmc_conv_summary.py gpurun_out/r06_pmc_conv_256x128 > gpurun_out/r06_pmc_conv_256x128.txt
bash tools/pmc_conv.sh "l1_out2b" gpurun_out/r06_pmc_conv_128x128 --cfgs 25 > /dev/null 2>&1
python tools/pmc_conv_summary.py gpurun_out/r06_pmc_conv_128x128 > gpurun_out/r06_pmc_conv_128x128.txt
find gpurun_out/r06_pmc_conv_256x128 gpurun_out/r06_pmc_conv_128x128 -name "*.csv" -size +2M -delete
cat gpurun_out/r06_pmc_conv_256x128.txt gpurun_out/r06_pmc_conv_128x128.txt
python tools/conv_bench.py --iters 20 > gpurun_out/r06_conv_bench.txt 2>&1
tail -60 gpurun_out/r06_conv_bench.txt
bash tools/gpu_tests.sh
