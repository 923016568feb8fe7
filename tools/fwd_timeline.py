"""Start / end of every kernel of ONE single-stream forward from a rocprofv3 kernel trace (csv): which stream ran what when.
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d out -o t -- python $REPO/bench.py --steps 3 --warmup 2 --images-per-step 2 --streams 1 --cpu-seconds 0 --no-legs --no-roofline
    python tools/fwd_timeline.py out/t_kernel_trace.csv"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last forward: from the last stem_direct kernel on
idx = [i for i, r in enumerate(rows) if "stem_direct" in r["Kernel_Name"]]
i0 = idx[-2] if len(idx) > 1 else idx[-1]
i1 = idx[-1] if len(idx) > 1 else len(rows)
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1]:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:56]
    print("%-58s q %3s  start %8.1f  dur %7.1f  grid %6s" % (n, r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3,
                                                            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
