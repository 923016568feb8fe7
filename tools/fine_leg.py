"""bench.py's fine leg on its own (full coarse-to-fine forward on the high-confidence fixture; dense fine map vs the match-driven
patch pyramid at the fixture's M and at ~300 matches, single stream and 3 streams):  python tools/fine_leg.py"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from onepose_plus_plus_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
print(json.dumps(bench.fine_leg(torch, dev, "bf16x3", _lib.load(), _lib), indent=1))
