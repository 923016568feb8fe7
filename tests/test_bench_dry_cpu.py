"""CPU: the multi-rank skeleton of bench.py (self-launch through torch.distributed.run, RANK / WORLD_SIZE handling, the ONE flat
weight broadcast, barrier + max-over-ranks timing, per-rank gather, rank 0's single JSON line) executed over gloo with a stand-in
for the forward (OPP_BENCH_DRY_RUN=1).  The RCCL runs of the same code need more than one GPU (tools/scale_check.sh)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [1, 2, 8])
def test_bench_skeleton_over_gloo(world):
    env = dict(os.environ, OPP_BENCH_DRY_RUN="1")
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1", "--images-per-step", "16"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                       # exactly ONE line, from rank 0
    assert len(lines[0]) < 6000                             # the driver keeps ~8 KB of stdout: the line must fit whole
    out = json.loads(lines[0])
    cfg = out["config"]
    # one node: LOCAL_RANK == RANK, and LOCAL_RANK is the device index run() hands torch.cuda.set_device (no HIP_VISIBLE_DEVICES games)
    assert [d["local_rank"] for d in cfg["rank_devices"]] == list(range(world))
    assert [d["device"] for d in cfg["rank_devices"]] == list(range(world))
    assert all(isinstance(d["host_affinity"], str) and d["host_affinity"] for d in cfg["rank_devices"])
    assert out["n_gpus"] == world and cfg["n_ranks_seen"] == world and len(cfg["rank_devices"]) == world
    assert [d["rank"] for d in cfg["rank_devices"]] == list(range(world))
    assert len({d["pid"] for d in cfg["rank_devices"]}) == world
    assert len({d["weights_checksum"] for d in cfg["rank_devices"]}) == 1          # rank 0's weights reached every rank
    pr = cfg["per_rank_images_per_s"]
    assert abs(pr["sum"] - sum(d["images_per_s"] for d in cfg["rank_devices"])) < 0.05
    total = 3 * 16 * world
    assert abs(out["value"] - total / (out["ms_per_step"] * 3 / 1e3)) / out["value"] < 1e-3    # value = whole job / max elapsed
    if world >= 2:
        rates = [d["images_per_s"] for d in cfg["rank_devices"]]
        assert rates[-1] < 0.8 * rates[0]                   # the stand-in makes rank r (1 + r / 2) x slower ...
        assert out["value"] < 1.1 * world * rates[-1]       # ... and the job is as fast as its slowest rank, not the sum


def _full_lines():
    import glob
    return sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_bench_driver_style*.json")) + glob.glob(os.path.join(ROOT, "profiles", "r0*_bench_detail*.json")))


@pytest.mark.parametrize("path", _full_lines(), ids=os.path.basename)
def test_compact_line_fits_the_driver_tail(path):
    """bench.py prints compact_line(everything it measured); fed with every full record committed under profiles/ (up to 20 KB, the
    record that the round-5 driver could not parse included) it must stay under 6000 bytes, round-trip through json and keep the
    contract's keys with `roofline` and `cpu_baseline` as numbers, not prose."""
    sys.path.insert(0, ROOT)
    import bench
    with open(path) as f:
        full = json.load(f)
    line = bench.bounded_dumps(bench.compact_line(full, "bench_detail.json"))
    assert len(line) < 6000 and "\n" not in line
    out = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in out, k
    assert out["value"] == full["value"] and out["ms_per_step"] == full["ms_per_step"]
    assert len(out["config"]["workload"]) <= 200
    r = out["roofline"]
    if full.get("roofline"):
        assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 2e-3
        assert r["traffic"] is None or isinstance(r["traffic"], int)
        assert len(r.get("top", [])) <= 8
    if full.get("cpu_baseline"):
        cb = out["cpu_baseline"]
        assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and len(cb["sample"]) <= 300


def test_compact_line_sheds_before_it_overflows():
    sys.path.insert(0, ROOT)
    import bench
    with open([p for p in _full_lines() if "rank_devices" in json.load(open(p))["config"]][-1]) as f:      # the newest FULL record (sidecar)
        full = json.load(f)
    full["config"]["rank_devices"] = [dict(full["config"]["rank_devices"][0], rank=i, local_rank=i, device=i) for i in range(8)]
    line = json.dumps(bench.compact_line(full, "bench_detail.json", limit=2000))
    assert len(line) <= 2000 and json.loads(line)["value"] == full["value"]


def test_executed_flops_exclude_the_cached_object_prefix():
    """337.3 GFLOP per image at N = 5000; with the per-object cache on, layer 0 on the 3D stream + its layer-1 QKV / KV
    work is not executed per image and is not counted (328.5)."""
    sys.path.insert(0, ROOT)
    import bench
    assert round(bench.flops_per_image(5000, False) / 1e9, 1) == 337.3
    assert round(bench.flops_per_image(5000, True) / 1e9, 1) == 328.5
    assert round(bench.flops_per_image(2000, False) / 1e9, 1) == 306.8 and round(bench.flops_per_image(15000, False) / 1e9, 1) == 438.8
