# kernel-by-kernel durations of the convolution backward bench (which part of a weight gradient is the wgrad kernel, which the reduce)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/wgtrace -o cb -- python $GRAFT_REPO_ROOT/tools/conv_bwd_bench.py 4 > $GRAFT_REPO_ROOT/gpurun_out/wgtrace.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
t = glob.glob("gpurun_out/wgtrace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(t)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = None
for r in rows:
    n = r["Kernel_Name"]
    if any(k in n for k in ("wgrad", "geo", "opp_gemm", "dilate", "flip", "splitk")):
        line = "%-70s %9.1f us  grid %s" % (n[:70], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000, r.get("Grid_Size", r.get("Grid_Size_X", "")))
        print(line)
PY
rm -f gpurun_out/wgtrace/*/*trace.csv gpurun_out/wgtrace/*trace.csv
