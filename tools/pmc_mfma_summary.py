"""MFMA-pipe utilisation per kernel launch shape from tools/pmc_mfma.sh.

  util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)

SQ_VALU_MFMA_BUSY_CYCLES sums, over all waves, the cycles a SIMD's matrix pipe is held by that wave's MFMAs (32 per
v_mfma_f32_32x32x16_bf16: checked against SQ_INSTS_MFMA); GRBM_GUI_ACTIVE comes back summed over the 8 XCDs; the chip has
256 CUs x 4 SIMDs = 1024 matrix pipes.  A bf16x3 product is six MFMAs, so util is the fraction of the chip's matrix-pipe cycles
spent issuing MFMAs at the clock the chip actually ran at (padding columns and recomputed halo included -- it bounds the
algorithmic fraction of bench.py's roofline from above).

    python tools/pmc_mfma_summary.py gpurun_out/pmc_mfma profiles/r03_pmc_mfma_util.csv
"""
import collections
import csv
import glob
import os
import sys


def main(d, out):
    files = glob.glob(os.path.join(d, "mfma", "**", "*counter_collection.csv"), recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            wg = int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"]))
            acc[(r["Kernel_Name"], wg)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    rows = []
    for (name, wg), c in acc.items():
        mean = {k: sum(v) / len(v) for k, v in c.items()}
        busy, gui = mean.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), mean.get("GRBM_GUI_ACTIVE", 0.0)
        if busy <= 0 or gui <= 0:
            continue
        n = len(c["SQ_VALU_MFMA_BUSY_CYCLES"])
        cycles = gui / 8.0
        rows.append((name, wg, n, mean.get("SQ_INSTS_MFMA", 0.0), busy, cycles, busy / (cycles * 1024.0)))
    rows.sort(key=lambda r: -r[4] * r[2])
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "WorkgroupsPerLaunch", "Launches", "MFMAInstructionsPerLaunch", "MFMABusyCyclesPerLaunch", "GfxCyclesPerLaunch",
                    "MFMAPipeUtilisation"])
        for r in rows:
            w.writerow([r[0], r[1], r[2], "%.0f" % r[3], "%.0f" % r[4], "%.0f" % r[5], "%.4f" % r[6]])
    for r in rows[:14]:
        print("%-70s wg %5d  n %3d  util %.3f" % (r[0][:70], r[1], r[2], r[6]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
