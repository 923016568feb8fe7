"""Phase breakdown (prologue / K loop / epilogue, shader clocks) of the two score sweeps (gemm_ss.hip) from in-kernel stamps.
    python -m onepose_plus_plus_amd.build --tuning
    OPP_HIP_LIB=onepose_plus_plus_amd/libopp_hip_tuning.so python tools/matcher_phases.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onepose_plus_plus_amd import OnePosePlus_model, default_config, _lib       # noqa: E402
from onepose_plus_plus_amd.synthetic import make_state_dict                      # noqa: E402


def main():
    cfg = default_config()
    dev = torch.device("cuda", 0)
    lib = _lib.load()
    N, hc, wc = 5000, 64, 64
    L = hc * wc
    g = torch.Generator().manual_seed(3)
    f2 = (torch.randn(L, 256, generator=g) * 4).to(dev)
    f3 = (torch.randn(N, 256, generator=g) * 4).to(dev)
    kpts = torch.rand(N, 3, device=dev)
    m = OnePosePlus_model(cfg).eval()
    m.load_state_dict(make_state_dict(cfg, 0), strict=True)
    m = m.to(dev)
    _, ctx = m._ensure_ready(dev)
    conf = torch.empty(1, N, L, device=dev)
    i_ids = torch.empty(N, dtype=torch.int64, device=dev)
    j_ids = torch.empty(N, dtype=torch.int64, device=dev)
    mconf = torch.empty(N, device=dev)
    mkc = torch.empty(N, 2, device=dev)
    mk3 = torch.empty(N, 3, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    nb = lib.opp_coarse_match_workspace_bytes(ctx, N, L)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    s = torch.cuda.current_stream(dev).cuda_stream
    tiles = -(-N // 128) * -(-L // 128)
    ts = torch.zeros(tiles * 4 * 4, dtype=torch.int64, device=dev)

    def run():
        _lib.check(lib.opp_coarse_match(ctx, f3.data_ptr(), f2.data_ptr(), N, hc, wc, kpts.data_ptr(), 8.0, None, conf.data_ptr(),
                                        i_ids.data_ptr(), j_ids.data_ptr(), mconf.data_ptr(), mkc.data_ptr(), mk3.data_ptr(),
                                        cnt.data_ptr(), ws.data_ptr(), nb, s), "coarse_match")
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    _lib.check(lib.opp_debug_timestamps(ts.data_ptr()), "ts")
    run()                      # both sweeps write the same stamp buffer: the second sweep's stamps survive
    torch.cuda.synchronize()
    _lib.check(lib.opp_debug_timestamps(None), "ts")
    t = ts.view(tiles, 4, 4).double().cpu()
    pro, loop, epi = (t[..., 1] - t[..., 0]), (t[..., 2] - t[..., 1]), (t[..., 3] - t[..., 2])
    span = t[..., 3].max() - t[..., 0].min()
    print("sweep (OPP_SS_TS_MODE=%s; 1 stats, 2 conf, unset: the second overwrites the first): tiles" % os.environ.get("OPP_SS_TS_MODE", "-"), "|", "tiles %d  prologue %.0f  loop %.0f (%.0f/chunk)  epilogue %.0f  | span %.0f clk, last tile starts at +%.0f"
          % (tiles, pro.mean(), loop.mean(), loop.mean() / 8, epi.mean(), span, t[..., 0].max() - t[..., 0].min()))
    order = t[:, 0, 0].argsort()
    st = t[order, 0, 0] - t[..., 0].min()
    print("tile start times (every 64th, clk):", [int(x) for x in st[::64]])
    print("per-tile total (clk): mean %.0f  min %.0f  max %.0f" % ((t[..., 3].amax(1) - t[..., 0].amin(1)).mean(), (t[..., 3].amax(1) - t[..., 0].amin(1)).min(), (t[..., 3].amax(1) - t[..., 0].amin(1)).max()))


if __name__ == "__main__":
    main()
