"""Micro-bench of the implicit-GEMM conv / dense GEMM kernel on the backbone / transformer
layer shapes (HIP-event timing through torch on the launch stream).

    python tools/conv_bench.py [--only NAME] [--iters 20]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from onepose_plus_plus_amd import _lib  # noqa: E402

# name, Hin, Win, cin, cout, ks, stride, cfgs
CONVS = [
    ("layer1 3x3 128->128 @256", 256, 256, 128, 128, 3, 1, [0, 10, 11, 20, 25, 106, 107, 101, 105, 102, 103]),
    ("l1_out2b 3x3 196->128 @256", 256, 256, 196, 128, 3, 1, [0, 10, 11, 20, 25]),
    ("l1_out2a 3x3 196->196 @256", 256, 256, 196, 196, 3, 1, [3, 11, 10, 0, 20, 22, 25, 27, 28]),
    ("l1_outconv 1x1 128->196 @256", 256, 256, 128, 196, 1, 1, [22, 25, 27]),
    # (27 = the 128 x 224 ring tile of the 196(->224)-column layers, bf16x3 only)
    # 196 = 192 + a 4-column tail: the 192-column part on the tuning library's 128 x 192 tile (config 140; run with
    # OPP_HIP_LIB=.../libopp_hip_tuning.so OPP_ABLATE=1, configs >= 100 are skipped otherwise)
    ("l1_out2a-192 3x3 196->192 @256", 256, 256, 196, 192, 3, 1, [140, 22, 25]),
    ("layer2-192 3x3 196->192 @128", 128, 128, 196, 192, 3, 1, [140, 25, 26]),
    ("layer2 3x3 196->196 @128", 128, 128, 196, 196, 3, 1, [5, 1, 2, 0, 22, 25, 26, 27]),
    ("layer2.0 3x3s2 128->196 @256", 256, 256, 128, 196, 3, 2, [5, 1, 2, 0, 22, 25, 26, 27]),
    ("l2_out2b 3x3 256->196 @128", 128, 128, 256, 196, 3, 1, [5, 1, 2, 0, 22, 25, 26, 27]),
    ("l2_out2a 3x3 256->256 @128", 128, 128, 256, 256, 3, 1, [1, 0, 2, 20, 22, 25, 26, 28]),
    ("layer3 3x3 256->256 @64", 64, 64, 256, 256, 3, 1, [2, 1, 26]),
]
# name, M, K, N, cfgs
DENSE = [
    ("dense 65536x1152x128", 65536, 1152, 128, [0, 20, 25]),
    ("dense 65536x128x128", 65536, 128, 128, [0]),
    ("qkv 9096x256x768", 9096, 256, 768, [0, 1, 20, 22, 25, 28]),
    ("qkv0 4096x256x768", 4096, 256, 768, [2, 20, 22, 25, 26]),          # layer 0's projection of the image stream alone (object prefix cached)
    ("merge 9096x256x256", 9096, 256, 256, [1, 2, 0, 25, 26]),
    ("mlp1 9096x512x512", 9096, 512, 512, [0, 1, 25, 26]),
    ("mlp2 9096x512x256", 9096, 512, 256, [1, 2, 0, 25, 26]),
    ("score 5000x256x4096", 5000, 256, 4096, [0, 20, 22]),
]


def pad32(c):
    return (c + 31) // 32 * 32


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


def split_w(lib, w, cfgs, prec, s):
    """pre-split the weight operand for `prec`; keep only tile configs built for it (+ the automatic choice)"""
    if prec == 1:
        w2 = torch.empty_like(w)
        _lib.check(lib.opp_pack_h2(w.data_ptr(), w2.data_ptr(), w.numel(), None, s), "pack_h2")
        return w2, sorted(set([c for c in cfgs if c in (0, 1, 2, 10, 11, 20, 22, 25, 26, 28, 30)] + [-1]))
    if prec == 2:
        w2 = torch.empty(w.numel() // 2 * 3, device="cuda")
        _lib.check(lib.opp_pack_b3(w.data_ptr(), w2.data_ptr(), w.numel(), s), "pack_b3")
        ok = (0, 1, 2, 10, 20, 22, 25, 26, 27, 30) + ((140,) if os.environ.get("OPP_ABLATE") else ())     # 140: the tuning library's 128 x 192 tile
        return w2, sorted(set([c for c in cfgs if c in ok] + [-1]))
    return w, cfgs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--cfgs", default=None, help="comma list: only these tile configs")
    ap.add_argument("--prec", type=int, default=2, help="operand arithmetic: 0 fp32, 1 fp16x2, 2 bf16x3 (default)")
    ap.add_argument("--split", type=int, default=0, help="bf16x3 convolutions through opp_conv2d_nhwc_split: 1 = pre-split input, fp32 output; "
                    "2 = pre-split input and output (no fp32 copy); 3 = pre-split input, both outputs")
    args = ap.parse_args()
    lib = _lib.load()
    s = torch.cuda.current_stream().cuda_stream
    for name, H, W, cin, cout, ks, stride, cfgs in CONVS:
        if args.only and args.only not in name:
            continue
        cip, cop = pad32(cin), pad32(cout)
        x = torch.randn(H, W, cip, device="cuda")
        w = torch.randn(cop, lib.opp_conv_packed_k(cin, ks), device="cuda") * 0.02
        Ho, Wo = H // stride, W // stride
        y = torch.empty(Ho, Wo, cop, device="cuda")
        bias = torch.randn(cop, device="cuda")
        alg = 2.0 * Ho * Wo * cout * ks * ks * cin
        padf = 2.0 * Ho * Wo * cop * ks * ks * cip
        w, cfgs = split_w(lib, w, cfgs, args.prec, s)
        if args.cfgs:
            cfgs = [int(c) for c in args.cfgs.split(",")]
        for cfg in cfgs:
            if cfg in (3, 4, 5, 13, 15) and cop % 224:
                continue
            if cfg >= 100 and not os.environ.get("OPP_ABLATE"):
                continue

            def fn():
                _lib.check(lib.opp_conv2d_nhwc(x.data_ptr(), H, W, cin, w.data_ptr(), bias.data_ptr(), cop, ks, stride,
                                               None, 0, 1, y.data_ptr(), cfg, args.prec, None, s), "conv")
            if args.split:
                if args.prec != 2 or (ks == 3 and cin % 32 != 0):
                    continue
                xs = torch.empty(x.numel() // 2 * 3, device="cuda")
                _lib.check(lib.opp_pack_b3(x.data_ptr(), xs.data_ptr(), x.numel(), s), "pack_b3")
                ys = torch.empty(y.numel() // 2 * 3, device="cuda")

                def fn():   # noqa: F811
                    _lib.check(lib.opp_conv2d_nhwc_split(None, xs.data_ptr(), H, W, cin, w.data_ptr(), bias.data_ptr(), cop, ks, stride, None, 0, 1,
                                                         y.data_ptr() if args.split != 2 else None, ys.data_ptr() if args.split >= 2 else None, cfg, s),
                               "conv_split")
            try:
                us = timeit(fn, args.iters)
                print("%-32s cfg%d  %8.1f us  alg %6.1f TF  padded %6.1f TF" % (name, cfg, us, alg / us / 1e6, padf / us / 1e6), flush=True)
            except Exception as e:
                print("%-32s cfg%d  FAILED %s" % (name, cfg, e), flush=True)
    for name, M, K, N, cfgs in DENSE:
        if args.only and args.only not in name:
            continue
        A = torch.randn(M, K, device="cuda")
        Wt = torch.randn(N, K, device="cuda")
        C = torch.empty(M, N, device="cuda")
        Wt, cfgs = split_w(lib, Wt, cfgs, args.prec, s)
        if args.cfgs:
            cfgs = [int(c) for c in args.cfgs.split(",")]
        for cfg in cfgs:
            def fn():
                _lib.check(lib.opp_linear(A.data_ptr(), M, K, Wt.data_ptr(), N, 0, C.data_ptr(), cfg, args.prec, None, s), "linear")
            try:
                us = timeit(fn, args.iters)
                print("%-32s cfg%d  %8.1f us  %6.1f TF" % (name, cfg, us, 2.0 * M * K * N / us / 1e6), flush=True)
            except Exception as e:
                print("%-32s cfg%d  FAILED %s" % (name, cfg, e), flush=True)


if __name__ == "__main__":
    main()
