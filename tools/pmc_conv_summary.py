"""Per-kernel averages of the counters collected by tools/pmc_conv.sh.
    python tools/pmc_conv_summary.py gpurun_out/pmc_h2"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(d, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "opp_gemm_kernel" not in k:
            continue
        short = k[k.find("<") + 1:k.find(">")] if "<" in k else k
        acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("   %-28s %14.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
