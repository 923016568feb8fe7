"""Per-launch durations of the conv tail and of the 192-column body from a rocprofv3 --kernel-trace CSV (one row per launch):
    python tools/tail_trace.py <dir with *kernel_trace.csv>"""
import csv
import glob
import sys
from collections import defaultdict

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        rows += list(csv.DictReader(fh))
by = defaultdict(list)
for r in rows:
    name = r.get("Kernel_Name", "")
    if "conv_tail_kernel" in name or "128, 192" in name or "128, 256, 2, 4, true" in name:
        short = name[name.find("conv_tail_kernel"):][:24] if "conv_tail_kernel" in name else name[name.find("opp_gemm_kernel"):][:48]
        key = (short, r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Grid_Size_Y"))
        by[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print("%-62s grid %-8s %-4s n=%-4d median %8.1f us  min %8.1f" % (k[0], k[1], k[2], len(v), v[len(v) // 2], v[0]))
