"""Per-kernel HBM traffic (bytes per launch) from the FETCH_SIZE / WRITE_SIZE passes of
tools/pmc_bench.sh.  FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B:
MI355X_MICROARCH.md §HBM); both counters are in KiB.

    python tools/pmc_traffic_summary.py gpurun_out/pmc_bench profiles/r02_pmc_hbm_traffic.csv profiles/traffic_symbols_bf16x3.json
"""
import collections
import re
import csv
import json
import os
import sys


def load(path):
    """(kernel name, workgroups per launch) -> counter values: one entry per LAUNCH SHAPE of a kernel symbol"""
    agg = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            wg = int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])) if "Grid_Size" in r and "Workgroup_Size" in r else 0
            agg[(r["Kernel_Name"], wg)].append(float(r["Counter_Value"]))
    return agg


def main(d, out):
    fetch = load(os.path.join(d, "FETCH_SIZE", "p_counter_collection.csv"))
    write = load(os.path.join(d, "WRITE_SIZE", "p_counter_collection.csv"))
    rows = []
    for k in fetch:
        f = sum(fetch[k]) / len(fetch[k]) * 1024 * 2
        w = sum(write.get(k, [0])) / max(1, len(write.get(k, [0]))) * 1024
        rows.append((k[0], len(fetch[k]), f, w, k[1]))
    rows.sort(key=lambda r: -(r[2] + r[3]) * r[1])
    with open(out, "w", newline="") as f:
        wr = csv.writer(f)
        wr.writerow(["Kernel", "WorkgroupsPerLaunch", "Launches", "FetchBytesPerLaunch(x2 corrected)", "WriteBytesPerLaunch"])
        for r in rows:
            wr.writerow([r[0], r[4], r[1], "%.0f" % r[2], "%.0f" % r[3]])
    # per kernel symbol -> bytes per launch, read by bench.py's roofline leg: GEMM kernels are keyed by their template
    # argument string, every other kernel by its bare function name
    sym = {}
    for k, n, fb, wb, wg in rows:
        if "opp_gemm_kernel<" in k:
            t = k[k.find("<") + 1:k.find(">")]
        elif "gemm_ss_res3_kernel" in k:      # the default single-sweep score GEMM since r06: profile symbol gemm_ss_kernel<3> (bench.py MFMA_SYMBOLS)
            t = "gemm_ss_kernel<3>"
        elif "gemm_ss_kernel<" in k:
            t = "gemm_ss_kernel<%s>" % k[k.find("<") + 1:k.find(">")]
        else:
            m = re.search(r"(?:::|^|\s)([A-Za-z_]\w*)\s*(?:<[^()]*>)?\(", k.replace("(anonymous namespace)", ""))
            t = m.group(1) if m else k
        t = "%s|grid %d" % (t, wg)          # one entry per launch shape (bench.py traffic_of)
        if t in sym:
            continue
        sym[t] = {"fetch_bytes_per_launch": round(fb), "write_bytes_per_launch": round(wb),
                  "hbm_bytes_per_launch": round(fb + wb), "launches_sampled": n,
                  "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over bench.py; FETCH x2 gfx950 correction"}
    jpath = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(out), "traffic_gemm_symbols.json")
    with open(jpath, "w") as f:
        json.dump(sym, f, indent=1)
    for t in sym:
        print(t, sym[t]["hbm_bytes_per_launch"])

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
