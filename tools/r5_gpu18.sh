set -x
cd $GRAFT_REPO_ROOT
bash tools/pmc_conv.sh "l1_out2a 3x3" gpurun_out/pmc224_256px --cfgs 22,27 > /dev/null 2>&1
python tools/pmc_conv_summary.py gpurun_out/pmc224_256px > gpurun_out/pmc224_256px.txt
bash tools/pmc_conv.sh "layer2 3x3 196" gpurun_out/pmc224_128px --cfgs 22,27,25 > /dev/null 2>&1
python tools/pmc_conv_summary.py gpurun_out/pmc224_128px > gpurun_out/pmc224_128px.txt
find gpurun_out/pmc224_256px gpurun_out/pmc224_128px -name "*trace.csv" -delete
cat gpurun_out/pmc224_256px.txt gpurun_out/pmc224_128px.txt
