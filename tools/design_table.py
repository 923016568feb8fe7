"""Markdown table of the per-symbol roofline for DESIGN.md from a bench line + the single-stream kernel trace + the PMC MFMA-utilisation summary:
    python tools/design_table.py gpurun_out/final_bench_driver.json gpurun_out/final_s1/s1_kernel_stats.csv profiles/r05_pmc_mfma_util.csv"""
import csv
import json
import sys

line = json.load(open(sys.argv[1]))
stats = {r["Name"]: r for r in csv.DictReader(open(sys.argv[2]))}
util = list(csv.DictReader(open(sys.argv[3]))) if len(sys.argv) > 3 else []


def trace_avg(sym):
    key = ("opp_gemm_kernel<%s>" % sym) if sym[0].isdigit() else sym.split("<")[0]
    for n, r in stats.items():
        if (key in n and ("gemm_ss" not in key or sym[-2] in n)) or (sym == "gemm_ss_kernel<3>" and "gemm_ss_res3_kernel" in n):
            return float(r["AverageNs"]) / 1e3, int(r["Calls"])
    return None, 0


def mfma_util(sym):
    key = ("opp_gemm_kernel<%s>" % sym) if sym[0].isdigit() else sym.split("<")[0]
    best = None
    for r in util:
        if (key in r["Kernel"] or (sym == "gemm_ss_kernel<3>" and "gemm_ss_res3_kernel" in r["Kernel"])) and \
                (best is None or int(r["Launches"]) > int(best["Launches"])):
            best = r
    return best["MFMAPipeUtilisation"] if best else "-"


roof = line["roofline"]
rows = [roof] + roof["other_kernels"]
print("| symbol | launches | µs / forward (events) | avg µs (events / trace) | achieved | frac (events / event-corrected) | MFMA-pipe busy (PMC) | HBM / launch (PMC) |")
print("|---|---|---|---|---|---|---|---|")
for m in rows:
    t, _ = trace_avg(m["symbol"])
    tr = m.get("traffic")
    hbm = "-"
    if isinstance(tr, dict):
        if "hbm_bytes_per_launch" in tr:
            hbm = "%.1f MB" % (tr["hbm_bytes_per_launch"] / 1e6)
        elif "per_launch_shape" in tr:
            hbm = " / ".join("%.1f" % (v["hbm_bytes_per_launch"] / 1e6) for v in tr["per_launch_shape"].values()) + " MB"
    unit = "TF" if m["unit"].startswith("TFLOP") else "GB/s"
    print("| %s | %g | %.0f | %.1f / %s | %.1f %s | %.3f / %.3f | %s | %s |" % (
        m["symbol"], m["launches_per_forward"], m["us_per_forward"], m["avg_launch_us"], ("%.1f" % t) if t else "-", m["achieved"], unit,
        m["frac"], m.get("frac_event_corrected", 0.0), mfma_util(m["symbol"]) if m["bound"] == "mfma" else "-", hbm))
c = line["config"]
print()
print("value %.1f images/s, ms_per_step %.3f, model_frac %.4f, model_tflops %.1f, policy %s, streams %d" % (
    line["value"], line["ms_per_step"], c["model_frac_of_mfma_peak"], c["model_tflops"], c["tile_policy"], c["streams_per_gpu"]))
print("event_pair_us", roof.get("event_pair_us"), "empty", roof.get("empty_kernel_us"), "extra", roof.get("event_extra_us"), roof.get("event_calibration"))
print(json.dumps(roof.get("legs"), indent=0)[:200])
