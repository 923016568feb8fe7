"""CPU: the fp16x2 operand split (oracle/fp16x2_oracle.py): representation error, weight scaling rule and the
error of the three-product scheme -- the numerics claims of DESIGN.md §4.1 that need no GPU."""
import numpy as np
import pytest

from oracle import fp16x2_oracle as X


def test_rtz_never_rounds_away_from_zero():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(200000) * np.exp(rng.uniform(-8, 8, 200000))).astype(np.float32)
    x = x[np.abs(x) < 6e4]
    h = X.fp16_rtz(x).astype(np.float32)
    assert (np.abs(h) <= np.abs(x)).all() and (np.sign(h) * np.sign(x) >= 0).all()
    ulp = np.maximum(np.abs(h) * 2.0 ** -10, 2.0 ** -24)
    assert (np.abs(x - h) < ulp * 1.0000001).all()


def test_split_carries_22_bits():
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(100000) * np.exp(rng.uniform(-4, 6, 100000))).astype(np.float32)
    x = x[(np.abs(x) > 2.0 ** -3) & (np.abs(x) < 6e4)]           # lo stays a normal fp16 number
    hi, lo = X.split_activations(x)
    rec = hi.astype(np.float64) + lo.astype(np.float64)
    assert (np.abs(rec - x) <= 2.0 ** -22 * np.abs(x)).all()
    tiny = (rng.standard_normal(1000) * 1e-3).astype(np.float32)  # below 2^-14 * 2^11: absolute, not relative
    hi, lo = X.split_activations(tiny)
    assert (np.abs(hi.astype(np.float64) + lo.astype(np.float64) - tiny) <= 2.0 ** -25).all()


@pytest.mark.parametrize("mag", [1e-6, 1e-4, 0.02, 1.0, 300.0, 3e4])
def test_weight_scale_rule(mag):
    rng = np.random.default_rng(2)
    w = (rng.standard_normal(4096) * mag).astype(np.float32)
    s, inv = X.weight_scale(w)
    assert s * inv == 1.0 and np.log2(float(s)) == np.round(np.log2(float(s)))       # exact power of two
    m = np.abs(w).max() * s
    assert 2.0 ** 14 <= m < 2.0 ** 15
    img, s2, _ = X.split_weights(w)
    hi = img[:, :8].reshape(-1).view(np.float16).astype(np.float64)
    lo = img[:, 8:].reshape(-1).view(np.float16).astype(np.float64)
    assert np.isfinite(hi).all() and np.isfinite(lo).all()
    big = np.abs(w) >= np.abs(w).max() * 2.0 ** -10                # entries within 2^-10 of the largest
    rec = (hi + lo) / float(s2)
    assert (np.abs(rec - w)[big] <= 2.0 ** -21 * np.abs(w)[big]).all()
    assert (np.abs(rec - w) <= 2.0 ** -21 * np.abs(w).max() * 2.0 ** -10 + 2.0 ** -21 * np.abs(w)).all()


def test_three_product_dot_error_is_fp32_class():
    rng = np.random.default_rng(3)
    A = rng.standard_normal((64, 1152)).astype(np.float32)
    W = (rng.standard_normal((32, 1152)) * 0.02).astype(np.float32)
    img, s, inv = X.split_weights(W)
    bh = img[:, :8].reshape(32, -1).view(np.float16)
    bl = img[:, 8:].reshape(32, -1).view(np.float16)
    got = X.dot_fp16x2(A, bh, bl) * float(inv)
    ref = A.astype(np.float64) @ W.astype(np.float64).T
    bound = np.abs(A).astype(np.float64) @ np.abs(W).astype(np.float64).T
    assert (np.abs(got - ref) / bound).max() < 2.0 ** -21       # operand error only: ~2^-22 per operand, random signs
