"""Stand-alone timing of the convolution backward kernels at the training step's shapes (B = 4, 512 x 512 images):
input gradient (implicit-GEMM kernel on the flipped weight) and weight gradient (conv_wgrad_kernel), TFLOP/s of the
algorithmic 2 * P * cout * cin * k * k each.     python tools/conv_bwd_bench.py [B] [--only <substring of the shape name>] [--what dgrad|wgrad] [--iters N]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from onepose_plus_plus_amd import _lib  # noqa: E402

SHAPES = [  # name, cin, cout, ks, stride, Hin (= Win)
    ("layer1 3x3 128->128 @256", 128, 128, 3, 1, 256),
    ("l1_out2a 3x3 196->196 @256", 196, 196, 3, 1, 256),
    ("l1_out2b 3x3 196->128 @256", 196, 128, 3, 1, 256),
    ("l1_out 1x1 128->196 @256", 128, 196, 1, 1, 256),
    ("layer2.0.conv1 3x3 s2 128->196 @256", 128, 196, 3, 2, 256),
    ("layer2 3x3 196->196 @128", 196, 196, 3, 1, 128),
    ("l2_out2a 3x3 256->256 @128", 256, 256, 3, 1, 128),
    ("layer3.0.conv1 3x3 s2 196->256 @128", 196, 256, 3, 2, 128),
    ("layer3 3x3 256->256 @64", 256, 256, 3, 1, 64),
]


def main():
    argv = sys.argv[1:]
    only = argv[argv.index("--only") + 1] if "--only" in argv else ""
    what_only = argv[argv.index("--what") + 1] if "--what" in argv else ""
    iters = int(argv[argv.index("--iters") + 1]) if "--iters" in argv else 200
    B = int(argv[0]) if argv and not argv[0].startswith("--") else 4
    lib = _lib.load()
    s = torch.cuda.current_stream().cuda_stream
    pad = lambda c: (c + 31) // 32 * 32
    for name, cin, cout, ks, stride, H in SHAPES:
        if only and only not in name:
            continue
        Ho = H // stride
        x = torch.randn(B, H, H, pad(cin), device="cuda")
        gy = torch.randn(B, Ho, Ho, pad(cout), device="cuda")
        gy[..., cout:] = 0
        w = torch.randn(cout, cin, ks, ks, device="cuda")
        gx = torch.empty_like(x)
        gw = torch.empty_like(w)
        nb = lib.opp_conv2d_backward_workspace_bytes(B, H, H, cin, cout, ks, stride, 2)
        ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
        flops = 2.0 * B * Ho * Ho * cout * cin * ks * ks
        res = []
        for what, a, b in (("dgrad", gx, None), ("wgrad", None, gw)):
            if what_only and what != what_only:
                continue

            def run():
                _lib.check(lib.opp_conv2d_backward_nhwc(x.data_ptr(), B, H, H, cin, w.data_ptr(), cout, ks, stride, gy.data_ptr(),
                                                        a.data_ptr() if a is not None else None, None, b.data_ptr() if b is not None else None, 2,
                                                        ws.data_ptr(), nb, s), "conv2d_backward")
            for _ in range(max(2, iters // 4)):     # the clock settles over tens of milliseconds: warm up with real work
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = iters
            e0.record()
            for _ in range(n):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            res.append("%s %.3f ms = %.0f TF" % (what, ms, flops / ms / 1e9))
        print("%-40s B=%d  %s" % (name, B, "   ".join(res)), flush=True)


if __name__ == "__main__":
    main()
