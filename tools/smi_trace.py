#!/usr/bin/env python3
"""Clock / power trace of the GPU while a command runs (the "effective clock" ceiling as evidence).

  python tools/smi_trace.py --out gpurun_out/smi -- python bench.py --steps 60 --warmup 5 --cpu-seconds 0 --no-legs

Two samplers next to the child process:
  * sysfs, every 50 ms: hwmon freq*_input (sclk / mclk, Hz), power1_average|power1_input (uW), temp1_input, and the
    starred line of pp_dpm_sclk -- whatever of these the driver exposes for the first amdgpu card;
  * `amd-smi metric --clock --power --usage --json` (falls back to `rocm-smi --showclocks --showpower --json`), once a second.
Writes <out>.sysfs.csv, <out>.smi.jsonl and <out>.summary.json (min / mean / max of every sampled quantity over the part of
the run in which the GPU was busy).  The amd-smi samples (`amd_smi` in the summary: gfx activity, socket power, per-XCD gfx
clocks of the one GPU the box exposes) are the authoritative ones; sysfs card0 is not necessarily the GPU in use on a multi-GPU
host.
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import threading
import time


def first_card():
    for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
        try:
            if open(os.path.join(dev, "vendor")).read().strip() == "0x1002":
                return dev
        except OSError:
            pass
    return None


def read(path):
    try:
        return open(path).read().strip()
    except OSError:
        return None


def sysfs_sources(dev):
    src = {}
    for hw in glob.glob(os.path.join(dev, "hwmon", "hwmon*")):
        for f in sorted(os.listdir(hw)):
            if f.startswith("freq") and f.endswith("_input"):
                label = read(os.path.join(hw, f.replace("_input", "_label"))) or f
                src["hz_" + label] = os.path.join(hw, f)
            elif f in ("power1_average", "power1_input"):
                src["uW_" + f] = os.path.join(hw, f)
            elif f == "temp1_input":
                src["mC_temp1"] = os.path.join(hw, f)
    return src


def starred_mhz(text):
    if not text:
        return None
    for line in text.splitlines():
        if line.rstrip().endswith("*"):
            try:
                return float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
            except (IndexError, ValueError):
                return None
    return None


def smi_rows(lines):
    """amd-smi metric --json samples -> [{t, gfx_activity_pct, socket_power_w, gfx_clk_mhz_mean / min / max over the XCDs}] of GPU 0
    (the one GPU the box exposes)"""
    out = []
    for ln in lines:
        try:
            rec = json.loads(ln)
            gpu = json.loads(rec["out"])
            gpu = (gpu.get("gpu_data") if isinstance(gpu, dict) else gpu)[0]
        except (ValueError, KeyError, IndexError, TypeError):
            continue
        row = {"t": round(rec.get("t", 0.0), 3)}
        try:
            row["gfx_activity_pct"] = float(gpu["usage"]["gfx_activity"]["value"])
        except (KeyError, TypeError, ValueError):
            pass
        try:
            row["socket_power_w"] = float(gpu["power"]["socket_power"]["value"])
        except (KeyError, TypeError, ValueError):
            pass
        clks = []
        for k, v in (gpu.get("clock") or {}).items():
            if k.startswith("gfx_"):
                try:
                    clks.append(float(v["clk"]["value"]))
                except (KeyError, TypeError, ValueError):
                    pass
        if clks:
            row.update({"gfx_clk_mhz_mean": sum(clks) / len(clks), "gfx_clk_mhz_min": min(clks), "gfx_clk_mhz_max": max(clks), "xcds": len(clks)})
            try:
                row["gfx_clk_mhz_limit"] = float(next(iter(gpu["clock"].values()))["max_clk"]["value"])
            except (KeyError, TypeError, ValueError, StopIteration):
                pass
        out.append(row)
    return out


def smi_summary(rows):
    busy = [r for r in rows if r.get("gfx_activity_pct", 0) >= 50]
    s = {"samples": len(rows), "busy_samples (gfx_activity >= 50 %)": len(busy)}
    for k in ("gfx_activity_pct", "socket_power_w", "gfx_clk_mhz_mean", "gfx_clk_mhz_min", "gfx_clk_mhz_max", "gfx_clk_mhz_limit"):
        v = [r[k] for r in busy if k in r]
        if v:
            s[k] = {"min": min(v), "mean": round(sum(v) / len(v), 1), "max": max(v)}
    return s


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--summarize":      # offline: tools/smi_trace.py --summarize <out>.smi.jsonl
        rows = smi_rows(open(sys.argv[2]).read().splitlines())
        print(json.dumps({"amd_smi": smi_summary(rows), "rows": rows}, indent=1))
        return 0
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--period", type=float, default=0.05)
    ap.add_argument("cmd", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    cmd = a.cmd[1:] if a.cmd and a.cmd[0] == "--" else a.cmd
    if not cmd:
        ap.error("no command")
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    dev = first_card()
    src = sysfs_sources(dev) if dev else {}
    dpm = os.path.join(dev, "pp_dpm_sclk") if dev else None
    stop = threading.Event()
    rows = []
    t0 = time.time()

    def sysfs_loop():
        while not stop.is_set():
            r = {"t": time.time() - t0}
            for k, p in src.items():
                v = read(p)
                try:
                    r[k] = float(v)
                except (TypeError, ValueError):
                    pass
            m = starred_mhz(read(dpm)) if dpm else None
            if m is not None:
                r["mhz_dpm_sclk"] = m
            rows.append(r)
            stop.wait(a.period)

    smi_lines = []

    def smi_loop():
        tools = [["amd-smi", "metric", "--clock", "--power", "--usage", "--json"],
                 ["rocm-smi", "--showclocks", "--showpower", "--json"]]
        tool = None
        while not stop.is_set():
            for t in ([tool] if tool else tools):
                try:
                    out = subprocess.run(t, capture_output=True, text=True, timeout=20)
                    if out.returncode == 0 and out.stdout.strip():
                        tool = t
                        smi_lines.append(json.dumps({"t": time.time() - t0, "tool": t[0], "out": out.stdout.strip()[:20000]}))
                        break
                except (OSError, subprocess.TimeoutExpired):
                    pass
            stop.wait(0.25)

    th = [threading.Thread(target=sysfs_loop, daemon=True), threading.Thread(target=smi_loop, daemon=True)]
    for t in th:
        t.start()
    rc = subprocess.call(cmd)
    stop.set()
    for t in th:
        t.join(timeout=30)

    keys = sorted({k for r in rows for k in r if k != "t"})
    with open(a.out + ".sysfs.csv", "w") as f:
        f.write(",".join(["t"] + keys) + "\n")
        for r in rows:
            f.write(",".join(["%.3f" % r["t"]] + ["%g" % r[k] if k in r else "" for k in keys]) + "\n")
    with open(a.out + ".smi.jsonl", "w") as f:
        f.write("\n".join(smi_lines) + ("\n" if smi_lines else ""))
    # busy window: power (or sclk) above the midpoint between its minimum and maximum over the run
    gate = next((k for k in keys if k.startswith("uW_")), None) or next((k for k in keys if k.startswith("hz_") or k.startswith("mhz_")), None)
    summary = {"command": cmd, "exit_code": rc, "seconds": time.time() - t0, "samples": len(rows), "card": dev, "gate": gate, "sources": src}
    if gate:
        vals = [r[gate] for r in rows if gate in r]
        mid = (min(vals) + max(vals)) / 2 if vals else 0
        busy = [r for r in rows if r.get(gate, 0) >= mid]
        summary["busy_samples"] = len(busy)
        for k in keys:
            v = [r[k] for r in busy if k in r]
            if v:
                summary[k] = {"min": min(v), "mean": sum(v) / len(v), "max": max(v)}
    srows = smi_rows(smi_lines)
    summary["amd_smi"] = smi_summary(srows)
    summary["amd_smi_rows"] = srows
    with open(a.out + ".summary.json", "w") as f:
        json.dump(summary, f, indent=1)
    print(json.dumps({k: v for k, v in summary.items() if k not in ("sources", "command", "amd_smi_rows")}))
    return rc


if __name__ == "__main__":
    sys.exit(main())
