"""CPU: the multi-rank skeleton of bench.py (self-launch through torch.distributed.run, RANK / WORLD_SIZE handling, the ONE flat
weight broadcast, barrier + max-over-ranks timing, per-rank gather, rank 0's single JSON line) executed over gloo with a stand-in
for the forward (OPP_BENCH_DRY_RUN=1).  The RCCL runs of the same code need more than one GPU (tools/scale_check.sh)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [1, 2])
def test_bench_skeleton_over_gloo(world):
    env = dict(os.environ, OPP_BENCH_DRY_RUN="1")
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1", "--images-per-step", "16"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                       # exactly ONE line, from rank 0
    out = json.loads(lines[0])
    cfg = out["config"]
    assert out["n_gpus"] == world and cfg["n_ranks_seen"] == world and len(cfg["rank_devices"]) == world
    assert [d["rank"] for d in cfg["rank_devices"]] == list(range(world))
    assert len({d["pid"] for d in cfg["rank_devices"]}) == world
    assert len({d["weights_checksum"] for d in cfg["rank_devices"]}) == 1          # rank 0's weights reached every rank
    pr = cfg["per_rank_images_per_s"]
    assert abs(pr["sum"] - sum(d["images_per_s"] for d in cfg["rank_devices"])) < 0.05
    total = 3 * 16 * world
    assert abs(out["value"] - total / (out["ms_per_step"] * 3 / 1e3)) / out["value"] < 1e-3    # value = whole job / max elapsed
    if world == 2:
        r0, r1 = (d["images_per_s"] for d in cfg["rank_devices"])
        assert r1 < 0.8 * r0                                # the stand-in makes rank 1 1.5x slower ...
        assert out["value"] < 2.2 * r1                      # ... and the job is as fast as its slowest rank, not the sum
