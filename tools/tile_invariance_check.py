"""Bit-invariance of the GEMM / conv results under the tile configuration (MI355X): every tile walks K in the same
order with the same product sequence, so the outputs must be identical, not just close.
    python tools/tile_invariance_check.py [--prec 3]"""
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, '.')
from tests import hip_ops as ops

prec = 3
g = torch.Generator().manual_seed(1)
bad = 0
for (M, K, N) in [(1000, 512, 512), (156, 128, 384), (4096, 64, 128), (300, 256, 768)]:
    A = torch.randn(M, K, generator=g); W = torch.randn(N, K, generator=g)
    base = None
    for cfg in ((25, 26, 22, 20, 2, 1, 0) if prec == 0 else (25, 26, 22, 20, 2)):
        out = ops.linear(A, W, 1, cfg, h2=prec)
        base = out if base is None else base
        ok = torch.equal(out, base)
        bad += not ok
        if not ok: print("dense", (M, K, N), "cfg", cfg, "DIFFERS", float((out - base).abs().max()))
cases = [(128, 128, 3, 1, 64, 64), (128, 196, 3, 2, 64, 64), (196, 196, 3, 1, 32, 48), (128, 196, 1, 2, 24, 40), (196, 256, 1, 1, 16, 24),
         (256, 196, 3, 1, 10, 12), (196, 128, 3, 1, 64, 64)]
for (cin, cout, ks, stride, H, Wd) in cases:
    x = torch.randn(1, cin, H, Wd, generator=g); w = torch.randn(cout, cin, ks, ks, generator=g) * 0.03
    scale = torch.rand(cout, generator=g) + 0.5; bias = torch.randn(cout, generator=g) * 0.1
    Ho, Wo = (H + 2 * (ks // 2) - ks) // stride + 1, (Wd + 2 * (ks // 2) - ks) // stride + 1
    res = torch.randn(1, cout, Ho, Wo, generator=g)
    for (rm, r, act) in [(0, None, 1), (1, res, 1), (0, None, 2)]:
        base = None
        for cfg in ((25, 26, 22, 20, 2, 1) if prec == 0 else (25, 26, 22, 20, 2) + ((27,) if (prec == 3 and cout == 196) else ())):
            out, _ = ops.conv2d(x, w, scale, bias, stride, r, rm, act, cfg, h2=prec)
            base = out if base is None else base
            ok = torch.equal(out, base)
            bad += not ok
            if not ok: print("conv", (cin, cout, ks, stride, H, Wd), "res", rm, "act", act, "cfg", cfg, "DIFFERS", float((out - base).abs().max()))
    if ks == 1 and stride == 1:
        low = torch.randn(1, cout, Ho // 2, Wo // 2, generator=g)
        base = None
        for cfg in ((25, 26, 22, 20, 2, 1) if prec == 0 else (25, 26, 22, 20, 2)):
            out, _ = ops.conv2d(x, w, None, None, 1, low, 2, 0, cfg, h2=prec)
            base = out if base is None else base
            ok = torch.equal(out, base)
            bad += not ok
            if not ok: print("conv1x1+bilinear cfg", cfg, "DIFFERS", float((out - base).abs().max()))
print("tile invariance:", "OK" if bad == 0 else "%d mismatches" % bad)
