"""Times the fine stage (window gather + loftr_fine + expectation head) at realistic match counts."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from onepose_plus_plus_amd import _lib, default_config  # noqa: E402
from onepose_plus_plus_amd.synthetic import make_state_dict, make_fine_ids  # noqa: E402
from tests import hip_ops as ops  # noqa: E402

cfg = default_config()
model = ops.make_model(cfg, make_state_dict(cfg, 0))
lib, ctx = ops.ctx_of(model)
N = 5000
feat = torch.randn(256, 256, 128, device="cuda")
bank = torch.randn(1, 128, N, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for M in (100, 500, 2000, 4000):
    i_ids, j_ids = make_fine_ids(N, (64, 64), M, 5)
    ii, jj = i_ids.cuda(), j_ids.cuda()
    mk = torch.zeros(M, 2, device="cuda")
    ex = torch.empty(M, 3, device="cuda")
    mf = torch.empty(M, 2, device="cuda")
    n = lib.opp_fine_workspace_bytes(ctx, M)
    ws = torch.empty(n, dtype=torch.uint8, device="cuda")

    def fn():
        _lib.check(lib.opp_fine(ctx, feat.data_ptr(), 256, 256, bank.data_ptr(), N, ii.data_ptr(), jj.data_ptr(), M, 64, 64,
                                mk.data_ptr(), 2.0, None, 1, ex.data_ptr(), mf.data_ptr(), ws.data_ptr(), n, s), "fine")
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    dt = (time.time() - t) / 20
    print("fine stage M=%5d: %.3f ms  (%.1f GFLOP -> %.1f TF)" % (M, dt * 1e3, M * 17.47e-3, M * 17.47e6 / dt / 1e12), flush=True)
