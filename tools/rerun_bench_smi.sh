# the two committed bench lines (default flags; the driver's --steps 20 --warmup 5) and a clock / power trace beside a longer run
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/final_bench_default.json 2> gpurun_out/final_bench_default.err
python bench.py --steps 20 --warmup 5 > gpurun_out/final_bench_driver.json 2> gpurun_out/final_bench_driver.err
if [ "$1" != "nosmi" ]; then
python tools/smi_trace.py --out gpurun_out/final_smi -- python bench.py --steps 200 --warmup 5 --cpu-seconds 0 --no-legs --no-roofline > gpurun_out/final_smi.log 2>&1
fi
