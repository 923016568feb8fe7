"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the bf16x3 operand split used by the default GEMM arithmetic
(DESIGN.md §4.1).  This is not a reference algorithm (the reference computes in fp32); it restates what
`b3_split_kernel` (csrc/backbone.hip) and the in-kernel activation split (csrc/gemm_mfma.hip `split_b3`) do, so
that the packed bytes can be checked bit-exactly and the exactness / error claims can be tested without a GPU.

    x  ->  hi  = bf16_rne(x)                 (v_cvt_pk_bf16_f32: round to nearest even)
           mid = bf16_rne(x - hi)            (x - hi is an exact fp32 subtraction)
           lo  = bf16_rne(x - hi - mid)      (exact again; <= 8 significant bits are left, so lo is exact too)
    hi + mid + lo == x  for every fp32 x with 2^-100 <~ |x| <= 3.39e38 (the largest bf16; lo must not underflow)
    a*b ~= lo_a*hi_b + hi_a*lo_b + mid_a*mid_b + mid_a*hi_b + hi_a*mid_b + hi_a*hi_b     (fp32 accumulate;
           the dropped mid*lo, lo*mid, lo*lo terms are <= 2^-26 |a b|)
"""
import numpy as np


def bf16_rne_bits(x):
    """float32 -> bfloat16 bit patterns (uint16), round to nearest even; NaN stays NaN (quiet)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    nan = (u & 0x7FFFFFFF) > 0x7F800000
    r = np.where(nan, (u >> 16) | 0x40, r)
    return r.astype(np.uint16)


def bf16_to_f32(b):
    return (b.astype(np.uint32) << 16).view(np.float32)


def split(x):
    """-> (hi, mid, lo) bf16 bit patterns (uint16 arrays shaped like x)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    with np.errstate(invalid="ignore", over="ignore"):
        hi = bf16_rne_bits(x)
        r1 = (x - bf16_to_f32(hi)).astype(np.float32)
        mid = bf16_rne_bits(r1)
        r2 = (r1 - bf16_to_f32(mid)).astype(np.float32)
        lo = bf16_rne_bits(r2)
    return hi, mid, lo


def split_packed(w):
    """K-contiguous fp32 matrix (size % 8 == 0) -> uint16 image [n/8][24] = [hi x8 | mid x8 | lo x8] per 8 values."""
    w = np.ascontiguousarray(w, dtype=np.float32).reshape(-1)
    assert w.size % 8 == 0
    hi, mid, lo = split(w)
    return np.concatenate([hi.reshape(-1, 8), mid.reshape(-1, 8), lo.reshape(-1, 8)], axis=1)


def dot_bf16x3(a, b):
    """Six-product dot products A[m,k] . B[n,k]^T (float64 accumulation stands in for the MFMA's fp32 accumulate)."""
    ah, am, al = (bf16_to_f32(t).astype(np.float64) for t in split(a))
    bh, bm, bl = (bf16_to_f32(t).astype(np.float64) for t in split(b))
    return al @ bh.T + ah @ bl.T + am @ bm.T + am @ bh.T + ah @ bm.T + ah @ bh.T
