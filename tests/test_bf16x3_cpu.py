"""CPU: the bf16x3 operand split (oracle/bf16x3_oracle.py) -- the numerics claims of DESIGN.md §4.1 about the default
GEMM arithmetic that need no GPU: the split is an exact (error-free) encoding of fp32 over its whole exponent range,
and the six kept products reproduce a*b to 2^-26."""
import numpy as np
import pytest

from oracle import bf16x3_oracle as X


def _wide(rng, n, lo_e=-60, hi_e=60):
    return (rng.standard_normal(n) * np.exp2(rng.uniform(lo_e, hi_e, n))).astype(np.float32)


def test_split_is_exact_over_the_fp32_range():
    rng = np.random.default_rng(0)
    x = np.concatenate([_wide(rng, 400000), np.float32([0.0, -0.0, 1.0, -1.0, 65504.0, 1e5, 1e-6, 3.38e38, -3.38e38,
                                                         1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24, 2.0 ** -90])])
    hi, mid, lo = (X.bf16_to_f32(t).astype(np.float64) for t in X.split(x))
    assert np.array_equal(hi + mid + lo, x.astype(np.float64))          # EXACT, not approximately
    ax = np.abs(x.astype(np.float64))
    assert (np.abs(mid) <= 2.0 ** -8 * ax).all() and (np.abs(lo) <= 2.0 ** -16 * ax).all()


def test_rne_matches_reference_rounding():
    rng = np.random.default_rng(1)
    x = _wide(rng, 100000, -20, 20)
    b = X.bf16_to_f32(X.bf16_rne_bits(x)).astype(np.float64)
    ulp = np.exp2(np.floor(np.log2(np.abs(x.astype(np.float64)))) - 7)
    assert (np.abs(b - x) <= 0.5 * ulp * (1 + 1e-12)).all()
    # ties go to the even mantissa
    t = np.float32([1.0 + 2.0 ** -8, 1.0 + 3 * 2.0 ** -8])
    assert X.bf16_to_f32(X.bf16_rne_bits(t)).tolist() == [1.0, 1.0 + 2.0 ** -6]


def test_nonfinite_inputs_stay_nonfinite():
    # (also the top 0.4 % of the fp32 range, above the largest bf16 3.39e38, whose hi rounds to infinity)
    hi, mid, lo = X.split(np.float32([np.inf, -np.inf, np.nan, 3.4e38]))
    rec = X.bf16_to_f32(hi) + X.bf16_to_f32(mid) + X.bf16_to_f32(lo)
    assert not np.isfinite(rec).any()


@pytest.mark.parametrize("sa,sb", [(1.0, 1.0), (1e5, 1e-6), (1e-6, 300.0), (3e4, 3e4)])
def test_six_products_reach_fp32_product_accuracy(sa, sb):
    """dropped terms <= 2^-26 |a||b| per product (round-to-nearest residuals: |mid| <= 2^-9, |lo| <= 2^-18 of |x|) --
    below the 2^-24 rounding of a single fp32 multiply-add, for any operand magnitude."""
    rng = np.random.default_rng(2)
    a = (rng.standard_normal((64, 256)) * sa).astype(np.float32)
    b = (rng.standard_normal((48, 256)) * sb).astype(np.float32)
    ref = a.astype(np.float64) @ b.astype(np.float64).T
    bound = np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64).T
    err = np.abs(X.dot_bf16x3(a, b) - ref) / bound
    assert err.max() <= 2.0 ** -26, err.max()


def test_packed_layout():
    w = np.arange(16, dtype=np.float32) * 1.000123 + 0.5
    img = X.split_packed(w)
    assert img.shape == (2, 24)
    hi, mid, lo = X.split(w)
    assert np.array_equal(img[1, :8], hi[8:]) and np.array_equal(img[1, 8:16], mid[8:]) and np.array_equal(img[1, 16:], lo[8:])
