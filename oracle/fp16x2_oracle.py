"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the fp16x2 operand split used by the default GEMM
arithmetic.  This is not a reference algorithm (the reference computes in fp32 / TF32);
it restates what `h2_scale_kernel` / `h2_split_kernel` (csrc/backbone.hip) and the in-kernel activation split
(csrc/gemm_mfma.hip `split_h2`) do, so that the packed bytes can be checked bit-exactly and the error bound of
the three-product scheme can be tested without a GPU.

    x  ->  hi = fp16_rtz(x)            (v_cvt_pkrtz_f16_f32: round toward zero)
           lo = fp16(x - hi)           weights: round toward zero again; activations: round to nearest even
    a*b ~= hi_a*lo_b + lo_a*hi_b + hi_a*hi_b      (fp32 accumulate; lo_a*lo_b <= 2^-22 |a b| is dropped)
"""
import numpy as np


def fp16_rtz(x):
    """float32 -> float16 rounding toward zero (finite inputs inside the fp16 range)."""
    x = np.asarray(x, dtype=np.float32)
    h = x.astype(np.float16)                      # round to nearest even
    hf = h.astype(np.float32)
    over = np.abs(hf) > np.abs(x)                 # rounded away from zero -> step one ulp back toward zero
    hb = h.view(np.uint16).copy()
    hb[over] -= 1                                 # sign-magnitude: decrementing the bit pattern shrinks |h|
    return hb.view(np.float16)


def weight_scale(w):
    """(s, 1/s): the power of two that brings max|w| into [2^14, 2^15) (h2_scale_kernel)."""
    m = float(np.max(np.abs(np.asarray(w, dtype=np.float32)[np.isfinite(w)]))) if np.size(w) else 0.0
    e = 14
    if m > 0.0:
        e = 14 - int(np.floor(np.log2(m)))
        if 2.0 ** (np.floor(np.log2(m))) > m:     # guard against log2 rounding at exact powers of two
            e += 1
    e = max(-100, min(100, e))
    return np.float32(2.0 ** e), np.float32(2.0 ** -e)


def split_weights(w, scaled=True):
    """K-contiguous fp32 matrix (size % 8 == 0) -> (uint16 image [n/8][16] = [hi x8 | lo x8] per 8 values, s, 1/s)."""
    w = np.ascontiguousarray(w, dtype=np.float32).reshape(-1)
    assert w.size % 8 == 0
    s, inv = weight_scale(w) if scaled else (np.float32(1.0), np.float32(1.0))
    v = (w * s).astype(np.float32)
    hi = fp16_rtz(v)
    lo = fp16_rtz((v - hi.astype(np.float32)).astype(np.float32))
    out = np.concatenate([hi.view(np.uint16).reshape(-1, 8), lo.view(np.uint16).reshape(-1, 8)], axis=1)
    return out, s, inv


def split_activations(x):
    """In-kernel split of the A operand: hi = rtz, lo = round-to-nearest of the exact remainder."""
    x = np.asarray(x, dtype=np.float32)
    hi = fp16_rtz(x)
    lo = (x - hi.astype(np.float32)).astype(np.float32).astype(np.float16)
    return hi, lo


def dot_fp16x2(a, b_hi, b_lo):
    """Three-product dot product (float64 accumulation stands in for the MFMA's fp32 accumulate)."""
    ah, al = split_activations(a)
    ah, al = ah.astype(np.float64), al.astype(np.float64)
    bh, bl = b_hi.astype(np.float64), b_lo.astype(np.float64)
    return ah @ bl.T + al @ bh.T + ah @ bh.T
