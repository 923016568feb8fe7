"""Per-dispatch counters of the gemm_ss kernel collected by tools/pmc_ss.sh (one line per probe shape).
    python tools/pmc_ss_summary.py gpurun_out/pmc_ss"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(d, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_ss_kernel" not in r["Kernel_Name"]:
            continue
        key = (int(r["Grid_Size"]) // 256)
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print("tiles", k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("   %-28s %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
