"""Loop efficiency of the split-operand LDS-DMA GEMM core (gemm_ss.hip) at several K (tuning build only).
    OPP_HIP_LIB=onepose_plus_plus_amd/libopp_hip_tuning.so python tools/gemm_ss_probe.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onepose_plus_plus_amd import _lib       # noqa: E402


def main():
    lib = _lib.load()
    fn = lib.opp_debug_gemm_ss_stats
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 2
    s = torch.cuda.current_stream().cuda_stream
    for M, N, K in ((4096, 5000, 256), (16384, 256, 2304), (65536, 128, 1152), (8192, 8192, 1024)):
        a = torch.randn(M, K, device="cuda")
        b = torch.randn(N, K, device="cuda")
        a3 = torch.empty(M * K // 2 * 3, device="cuda")
        b3 = torch.empty(N * K // 2 * 3, device="cuda")
        _lib.check(lib.opp_pack_b3(a.data_ptr(), a3.data_ptr(), a.numel(), s), "pack")
        _lib.check(lib.opp_pack_b3(b.data_ptr(), b3.data_ptr(), b.numel(), s), "pack")
        tm, tn = -(-M // 128), -(-N // 128)
        stats = torch.empty(2 * (M * tn + tm * N) + 64, device="cuda")
        tiles = tm * tn
        ts = torch.zeros(tiles * 4 * 4, dtype=torch.int64, device="cuda")
        for _ in range(2):
            _lib.check(fn(a3.data_ptr(), b3.data_ptr(), M, N, K, stats.data_ptr(), s), "gemm_ss")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            _lib.check(fn(a3.data_ptr(), b3.data_ptr(), M, N, K, stats.data_ptr(), s), "gemm_ss")
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 5
        _lib.check(lib.opp_debug_timestamps(ts.data_ptr()), "ts")
        _lib.check(fn(a3.data_ptr(), b3.data_ptr(), M, N, K, stats.data_ptr(), s), "gemm_ss")
        torch.cuda.synchronize()
        _lib.check(lib.opp_debug_timestamps(None), "ts")
        t = ts.view(tiles, 4, 4).double().cpu()
        pro, loop, epi = (t[..., 1] - t[..., 0]), (t[..., 2] - t[..., 1]), (t[..., 3] - t[..., 2])
        tf = 2.0 * M * N * K / us * 1e-6
        print("M %6d N %5d K %5d tiles %5d : %8.1f us  %6.1f TFLOP/s (%.3f of 416.7) | prologue %6.0f  loop %8.0f (%5.0f / k16-stage; MFMA floor 768 alone, 1536 "
              "shared)  epilogue %6.0f clk" % (M, N, K, tiles, us, tf, tf / 416.7, pro.mean(), loop.mean(), loop.mean() / (K / 16), epi.mean()), flush=True)


if __name__ == "__main__":
    main()
