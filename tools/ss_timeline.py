"""Per-CU timeline of the score GEMM (gemm_ss.hip) from in-kernel stamps: which workgroups shared a CU, when each tile's prologue / K loop /
epilogue ran, how much of a K loop overlapped the partner's K loop.
    python -m onepose_plus_plus_amd.build --tuning
    OPP_HIP_LIB=onepose_plus_plus_amd/libopp_hip_tuning.so [OPP_SS_PERSIST=0|1] [OPP_SS_DELAY=cycles] python tools/ss_timeline.py"""
import os
import sys
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onepose_plus_plus_amd import OnePosePlus_model, default_config, _lib       # noqa: E402
from onepose_plus_plus_amd.synthetic import make_state_dict                      # noqa: E402


def main():
    cfg = default_config()
    dev = torch.device("cuda", 0)
    lib = _lib.load()
    N, hc, wc = int(os.environ.get("N", "5000")), 64, 64
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
    run()
    torch.cuda.synchronize()
    _lib.check(lib.opp_debug_timestamps(None), "ts")
    t = ts.view(tiles, 4, 4).cpu()
    persist = os.environ.get("OPP_SS_PERSIST", "1") != "0"
    w0 = t[:, 0, :].double()                      # wave 0: start, prologue done, loop done, end
    base = w0[:, 0].min()
    w0 = w0 - base
    pro, loop, epi = w0[:, 1] - w0[:, 0], w0[:, 2] - w0[:, 1], w0[:, 3] - w0[:, 2]
    print("%s kernel, %d tiles: prologue %.0f  loop %.0f (%.0f / k16-stage)  epilogue %.0f  | span %.0f clk"
          % ("persistent" if persist else "one-tile", tiles, pro.mean(), loop.mean(), loop.mean() / 16, epi.mean(), w0[:, 3].max()))
    if not persist:
        return
    hw = t[:, 1, 0]
    xcc = (hw >> 32) & 0xf
    hwid = hw & 0xffffffff
    # gfx9 HW_ID: wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh_id [12], se_id [15:13] (se bits differ per chip: keep everything above bit 8)
    cu_key = [(int(x), int(h) >> 8 & 0xff) for x, h in zip(xcc.tolist(), hwid.tolist())]
    slot = [int(h) & 0xf for h in hwid.tolist()]
    per_cu = defaultdict(list)
    for i in range(tiles):
        per_cu[cu_key[i]].append((float(w0[i, 0]), float(w0[i, 1]), float(w0[i, 2]), float(w0[i, 3]), slot[i], i))
    ncu = len(per_cu)
    counts = sorted(len(v) for v in per_cu.values())
    print("CUs seen %d, tiles per CU min %d max %d; first-tile start of the late residents: %.0f clk"
          % (ncu, counts[0], counts[-1], sorted(v[0] for vs in per_cu.values() for v in vs)[min(tiles - 1, 256 + 8)]))
    # overlap of K loops on a CU: fraction of a loop's duration during which another tile of the same CU is also in its loop
    ov_tot, loop_tot, end_cu = 0.0, 0.0, []
    for vs in per_cu.values():
        end_cu.append(max(v[3] for v in vs))
        for a in vs:
            la = a[2] - a[1]
            loop_tot += la
            for b in vs:
                if b is a:
                    continue
                ov_tot += max(0.0, min(a[2], b[2]) - max(a[1], b[1]))
    print("K-loop time overlapped by the partner's K loop: %.1f %%; CU finish time mean %.0f max %.0f clk" % (100 * ov_tot / loop_tot, sum(end_cu) / ncu, max(end_cu)))
    # by position in the workgroup's tile list
    order = defaultdict(list)
    for vs in per_cu.values():
        by_slot = defaultdict(list)
        for v in vs:
            by_slot[v[4]].append(v)
        for sl, lst in by_slot.items():
            lst.sort()
            for k, v in enumerate(lst):
                order[k].append(v)
    for k in sorted(order):
        lst = order[k]
        n = len(lst)
        print("  tile #%d of its workgroup (%4d tiles): start %.0f  prologue %.0f  loop %.0f  epilogue %.0f"
              % (k, n, sum(v[0] for v in lst) / n, sum(v[1] - v[0] for v in lst) / n, sum(v[2] - v[1] for v in lst) / n, sum(v[3] - v[2] for v in lst) / n))
    one = next(iter(per_cu.values()))
    print("one CU:", [(int(v[4]), int(v[0]), int(v[1]), int(v[2]), int(v[3])) for v in sorted(one)])


if __name__ == "__main__":
    main()
