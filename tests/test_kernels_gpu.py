"""GPU: building-block kernels (fp32 MFMA GEMM / implicit-GEMM conv, LayerNorm) against a
float64 CPU evaluation of the same op.  Called through the C ABI."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _needs_fp16x2():
    from tests import hip_ops as ops
    if not ops.has_fp16x2():
        pytest.skip("fp16x2 kernels are built into the tuning library only (round 6)")


def _bound(A, W):
    return (A.abs().double() @ W.abs().double().T).float()


@pytest.mark.parametrize("M,K,N", [(300, 256, 768), (130, 512, 256), (64, 128, 128), (1000, 64, 128), (77, 256, 96),
                                   (50, 64, 81), (33, 96, 7)])
@pytest.mark.parametrize("cfg", [-1, 0, 1, 2, 20, 22, 25, 26])
def test_linear_tiles(M, K, N, cfg):
    from tests import hip_ops as ops
    g = torch.Generator().manual_seed(M + K + N)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g)   # asymmetric: catches transposed operands / outputs
    ref = (A.double() @ W.double().T).float()
    got = ops.linear(A, W, 0, cfg)
    assert torch.isfinite(got).all()
    err = (got - ref).abs()
    assert (err <= 2e-6 * _bound(A, W) + 1e-6).all(), float(err.max())


@pytest.mark.parametrize("M,K,N", [(300, 256, 768), (130, 512, 256), (64, 128, 128), (77, 256, 96), (50, 64, 81),
                                   (33, 96, 7), (515, 1152, 130)])
@pytest.mark.parametrize("cfg", [-1, 2, 20, 22, 25, 26])
def test_linear_fp16x2(M, K, N, cfg):
    """fp16x2-split operands (hi + lo fp16, three fp16 MFMAs, fp32 accumulate): same error class as fp32."""
    _needs_fp16x2()
    from tests import hip_ops as ops
    g = torch.Generator().manual_seed(M + K + N)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g)
    ref = (A.double() @ W.double().T).float()
    got = ops.linear(A, W, 0, cfg, h2=1)
    assert torch.isfinite(got).all()
    err = (got - ref).abs()
    assert (err <= 2e-6 * _bound(A, W) + 1e-6).all(), float(err.max())


@pytest.mark.parametrize("M,K,N", [(300, 256, 768), (130, 512, 256), (64, 128, 128), (77, 256, 96), (50, 64, 81),
                                   (33, 96, 7), (515, 1152, 130), (1000, 64, 128)])
@pytest.mark.parametrize("cfg", [-1, 2, 20, 22, 25, 26])
def test_linear_bf16x3(M, K, N, cfg):
    """bf16x3-split operands (exact hi + mid + lo bf16, six bf16 MFMAs, fp32 accumulate): the default arithmetic."""
    from tests import hip_ops as ops
    g = torch.Generator().manual_seed(M + K + N)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g)
    ref = (A.double() @ W.double().T).float()
    got = ops.linear(A, W, 0, cfg, h2=3)
    assert torch.isfinite(got).all()
    err = (got - ref).abs()
    assert (err <= 1e-6 * _bound(A, W) + 1e-6).all(), float(err.max())
    for bad in (28, 29, 3):      # tiles without a bf16x3 build are refused, never silently another arithmetic
        with pytest.raises(Exception):
            ops.linear(A, W, 0, bad, h2=3)


def test_linear_bf16x3_identity_layout():
    """A = I with an asymmetric W: the six products reproduce W^T EXACTLY (the split is an error-free encoding)."""
    from tests import hip_ops as ops
    g = torch.Generator().manual_seed(11)
    W = torch.randn(128, 64, generator=g) * torch.exp2(torch.randint(-30, 30, (128, 1), generator=g).float())
    got = ops.linear(torch.eye(64), W, 0, -1, h2=3)
    assert torch.equal(got, W.T.contiguous())


@pytest.mark.parametrize("sa,sw", [(1.0, 1.0), (1e5, 0.02), (1e-6, 1.0), (3e4, 300.0), (1e-6, 1e-6), (1e10, 1e-10)])
def test_bf16x3_not_narrower_than_fp32(sa, sw):
    """Error of the bf16x3 GEMM vs fp64, relative to sum|a||w|, against the exact-fp32 MFMA GEMM of the same inputs:
    no larger (both are fp32-accumulation round-off), at ANY operand magnitude -- activations of 1e5 and 1e-6
    included, where the fp16x2 fast mode overflows / loses relative precision."""
    from tests import hip_ops as ops
    g = torch.Generator().manual_seed(5)
    A = torch.randn(256, 1152, generator=g) * sa
    W = torch.randn(192, 1152, generator=g) * sw
    ref = A.double() @ W.double().T
    bound = A.abs().double() @ W.abs().double().T
    e_b3 = ((ops.linear(A, W, 0, -1, h2=3).double() - ref).abs() / bound)
    e_f32 = ((ops.linear(A, W, 0, -1, h2=0).double() - ref).abs() / bound)
    print("bf16x3 rel err max %.3e mean %.3e | fp32 MFMA max %.3e mean %.3e" %
          (e_b3.max(), e_b3.mean(), e_f32.max(), e_f32.mean()))
    assert e_b3.max() <= 1.25 * e_f32.max() + 2.0 ** -26, (float(e_b3.max()), float(e_f32.max()))
    assert e_b3.mean() <= 1.25 * e_f32.mean() + 2.0 ** -28
    if sa == 1e-6 and sw == 1.0 and ops.has_fp16x2():    # (tuning library) what the exactness buys: fp16x2 is absolute-error limited down there
        e_h2 = ((ops.linear(A, W, 0, -1, h2=1).double() - ref).abs() / bound).max().item()
        assert e_h2 > 100 * e_b3.max().item(), (e_h2, float(e_b3.max()))


@pytest.mark.parametrize("mag", [1e-30, 1e-5, 1.0, 700.0, 1e20])
def test_pack_b3_bit_exact_vs_oracle(mag):
    """`opp_pack_b3` (operand pre-split for the bf16x3 GEMMs): byte-exact against oracle/bf16x3_oracle.py."""
    import numpy as np
    from oracle import bf16x3_oracle as X
    from tests import hip_ops as ops
    g = torch.Generator().manual_seed(3)
    w = torch.randn(192, 1152, generator=g) * mag
    w[3, 5] = 0.0
    w[7, :8] = torch.tensor([mag * 2.0 ** -12, -mag * 2.0 ** -20, mag * (1 + 2.0 ** -8), -mag * (1 + 3 * 2.0 ** -8), mag, -mag,
                             mag / 3, -mag / 7])
    out = ops.pack_b3(w.cuda())
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint16).reshape(-1, 24)
    assert np.array_equal(got, X.split_packed(w.numpy()))


@pytest.mark.parametrize("sa,sw", [(1.0, 0.02), (1.0, 1e-4), (1.0, 300.0), (30.0, 1e-6), (0.05, 0.02)])
def test_linear_fp16x2_weight_magnitudes(sa, sw):
    """The per-matrix power-of-two weight scale keeps the fp16 lo halves normal: accuracy does not depend on
    the weight magnitude; the un-scaled split (h2=2) degrades for small weights (fp16 subnormal lo)."""
    _needs_fp16x2()
    from tests import hip_ops as ops
    g = torch.Generator().manual_seed(5)
    A = torch.randn(256, 1152, generator=g) * sa
    W = torch.randn(192, 1152, generator=g) * sw
    ref = A.double() @ W.double().T
    bound = A.abs().double() @ W.abs().double().T
    e_scaled = ((ops.linear(A, W, 0, -1, h2=1).double() - ref).abs() / bound).max().item()
    assert e_scaled < 1e-6, e_scaled
    if sw <= 1e-4:
        e_raw = ((ops.linear(A, W, 0, -1, h2=2).double() - ref).abs() / bound).max().item()
        assert e_raw > 4 * e_scaled, (e_raw, e_scaled)


@pytest.mark.parametrize("cfg", [10, 11, -1])
@pytest.mark.parametrize("N", [224, 448])
def test_linear_224_tiles(cfg, N):
    from tests import hip_ops as ops
    g = torch.Generator().manual_seed(N)
    A = torch.randn(333, 96, generator=g)
    W = torch.randn(N, 96, generator=g)
    ref = (A.double() @ W.double().T).float()
    got = ops.linear(A, W, 1, cfg)
    assert ((got - ref.clamp(min=0)).abs() <= 2e-6 * _bound(A, W) + 1e-6).all()


def test_linear_identity_layout():
    """A = I with an asymmetric W: output must equal W^T exactly (row/col swap detector)."""
    from tests import hip_ops as ops
    W = torch.arange(128 * 64, dtype=torch.float32).reshape(128, 64) / 7.0
    A = torch.eye(64)
    got = ops.linear(A, W, 0, -1)
    assert torch.equal(got, W.T.contiguous())


CONV_CASES = [
    # cin, cout, ks, stride, H, W
    (128, 128, 3, 1, 24, 40),
    (128, 196, 3, 2, 24, 40),
    (196, 196, 3, 1, 16, 24),
    (128, 196, 1, 2, 24, 40),
    (196, 256, 1, 1, 16, 24),
    (256, 196, 3, 1, 10, 12),
    (256, 256, 3, 1, 8, 8),
    # K tail packing (3x3 over 32 n + 1..4 channels: the last channels of all taps share two chunks)
    (196, 128, 3, 2, 17, 23),
    (68, 64, 3, 1, 9, 33),
    (97, 64, 3, 1, 12, 12),
    (101, 64, 3, 1, 12, 12),      # 5 tail channels: NOT packed
]


@pytest.mark.parametrize("cin,cout,ks,stride,H,W", CONV_CASES)
@pytest.mark.parametrize("cfg", [-1, 0, 2, 1])
def test_conv_vs_torch(cin, cout, ks, stride, H, W, cfg):
    from tests import hip_ops as ops
    if cout == 196 and cfg != -1:
        cfg = {0: 10, 2: 2, 1: 11}[cfg]
    g = torch.Generator().manual_seed(cin * 7 + cout + ks + stride)
    x = torch.randn(1, cin, H, W, generator=g)
    w = torch.randn(cout, cin, ks, ks, generator=g) * (2.0 / (cin * ks * ks)) ** 0.5
    scale = torch.rand(cout, generator=g) + 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    Ho, Wo = (H + 2 * (ks // 2) - ks) // stride + 1, (W + 2 * (ks // 2) - ks) // stride + 1
    res = torch.randn(1, cout, Ho, Wo, generator=g)
    ref = F.conv2d(x.double(), (w * scale.view(-1, 1, 1, 1)).double(), bias.double(), stride, ks // 2) + res.double()
    ref = F.relu(ref).float()
    got, pad_max = ops.conv2d(x, w, scale, bias, stride, res, 1, 1, cfg)
    assert pad_max == 0.0          # padded channels must stay exactly zero
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max() <= 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("cin,cout,ks,stride,H,W", CONV_CASES)
@pytest.mark.parametrize("cfg", [-1, 2, 20, 22, 25, 26])
def test_conv_fp16x2(cin, cout, ks, stride, H, W, cfg):
    _needs_fp16x2()
    from tests import hip_ops as ops
    g = torch.Generator().manual_seed(cin * 7 + cout + ks + stride)
    x = torch.randn(1, cin, H, W, generator=g)
    w = torch.randn(cout, cin, ks, ks, generator=g) * (2.0 / (cin * ks * ks)) ** 0.5
    scale = torch.rand(cout, generator=g) + 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    Ho, Wo = (H + 2 * (ks // 2) - ks) // stride + 1, (W + 2 * (ks // 2) - ks) // stride + 1
    res = torch.randn(1, cout, Ho, Wo, generator=g)
    ref = F.conv2d(x.double(), (w * scale.view(-1, 1, 1, 1)).double(), bias.double(), stride, ks // 2) + res.double()
    ref = F.relu(ref).float()
    got, pad_max = ops.conv2d(x, w, scale, bias, stride, res, 1, 1, cfg, h2=1)
    assert pad_max == 0.0
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max() <= 2e-5 * max(1.0, ref.abs().max().item())
    with pytest.raises(Exception):          # the 224-column tiles have no fp16x2 build: refused, not silently fp32
        ops.conv2d(x, w, scale, bias, stride, res, 1, 1, 3, h2=1)


@pytest.mark.parametrize("cin,cout,ks,stride,H,W", CONV_CASES)
@pytest.mark.parametrize("cfg", [-1, 2, 20, 22, 25, 26])
def test_conv_bf16x3(cin, cout, ks, stride, H, W, cfg):
    from tests import hip_ops as ops
    g = torch.Generator().manual_seed(cin * 7 + cout + ks + stride)
    x = torch.randn(1, cin, H, W, generator=g)
    w = torch.randn(cout, cin, ks, ks, generator=g) * (2.0 / (cin * ks * ks)) ** 0.5
    scale = torch.rand(cout, generator=g) + 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    Ho, Wo = (H + 2 * (ks // 2) - ks) // stride + 1, (W + 2 * (ks // 2) - ks) // stride + 1
    res = torch.randn(1, cout, Ho, Wo, generator=g)
    ref = F.conv2d(x.double(), (w * scale.view(-1, 1, 1, 1)).double(), bias.double(), stride, ks // 2) + res.double()
    ref = F.relu(ref).float()
    got, pad_max = ops.conv2d(x, w, scale, bias, stride, res, 1, 1, cfg, h2=3)
    assert pad_max == 0.0
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max() <= 1e-5 * max(1.0, ref.abs().max().item())
    with pytest.raises(Exception):          # the 224-column tiles have no bf16x3 build: refused, not silently fp32
        ops.conv2d(x, w, scale, bias, stride, res, 1, 1, 3, h2=3)


# pre-split activations (round 6): cin, cout, ks, stride, H, W, res_mode, act
PRESPLIT_CASES = [
    (128, 128, 3, 1, 40, 36, 1, 1),     # layer1: shortcut + ReLU
    (128, 196, 3, 2, 40, 36, 0, 1),     # layer2.0.conv1, stride 2, 224 stored columns
    (128, 196, 1, 2, 40, 36, 0, 0),     # layer2.0.downsample
    (196, 256, 1, 2, 24, 20, 0, 0),     # layer3.0.downsample: 1 x 1 over 196 (224 stored) channels -- no packed K tail
    (196, 256, 1, 1, 24, 20, 2, 0),     # layer2_outconv + bilinear x2 residual
    (256, 256, 3, 1, 16, 16, 1, 1),     # layer3 (few tiles under a long K: runs as K slices + the fixed-order epilogue)
    (256, 196, 3, 1, 24, 20, 0, 2),     # layer2_outconv2[3], LeakyReLU upstream
    (64, 32, 3, 1, 9, 7, 0, 0),         # two channel groups, ragged tile
    (32, 64, 1, 1, 5, 3, 0, 0),         # a single chunk
]


@pytest.mark.parametrize("cin,cout,ks,stride,H,W,res_mode,act", PRESPLIT_CASES)
@pytest.mark.parametrize("cfg", [-1, 2, 20, 22, 25, 26])
def test_conv_presplit_is_bit_identical(cin, cout, ks, stride, H, W, res_mode, act, cfg):
    """opp_conv2d_nhwc_split (input read as the bf16 triples its producer wrote, output written as triples) against opp_conv2d_nhwc(prec=2),
    which splits the same fp32 values in its K loop: the same bits in, the same MFMA sequence -> the same bits out, on every tile; the
    pre-split output rows are exactly opp_pack_b3 of the fp32 rows."""
    from tests import hip_ops as ops
    g = torch.Generator().manual_seed(cin * 7 + cout + ks + stride + H)
    x = torch.randn(1, cin, H, W, generator=g)
    w = torch.randn(cout, cin, ks, ks, generator=g) * (2.0 / (cin * ks * ks)) ** 0.5
    scale = torch.rand(cout, generator=g) + 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    Ho, Wo = (H + 2 * (ks // 2) - ks) // stride + 1, (W + 2 * (ks // 2) - ks) // stride + 1
    res = None
    if res_mode == 1:
        res = torch.randn(1, cout, Ho, Wo, generator=g)
    elif res_mode == 2:
        res = torch.randn(1, cout, Ho // 2, Wo // 2, generator=g)
    y, ys, ref, ref_s = ops.conv2d_split(x, w, scale, bias, stride, res, res_mode, act, cfg, give_fp32=False)
    assert torch.isfinite(ref).all()
    assert torch.equal(y, ref)
    assert torch.equal(ys.view(torch.int32), ref_s.view(torch.int32))
    # the split rows alone (no fp32 copy of the output), input offered both ways
    y2, ys2, _, _ = ops.conv2d_split(x, w, scale, bias, stride, res, res_mode, act, cfg, want_fp32=False)
    assert y2 is None and torch.equal(ys2.view(torch.int32), ref_s.view(torch.int32))


@pytest.mark.parametrize("cfg", [-1, 22, 25])
def test_conv_presplit_falls_back_to_fp32_rows_for_a_packed_k_tail(cfg):
    """3 x 3 over 196 channels packs its last 4 channels 8 taps to a chunk: that K walk reads the fp32 rows; the split OUTPUT is still written.
    Without the fp32 rows the call is refused."""
    from tests import hip_ops as ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 196, 20, 24, generator=g)
    w = torch.randn(196, 196, 3, 3, generator=g) * 0.03
    y, ys, ref, ref_s = ops.conv2d_split(x, w, None, None, 1, None, 0, 1, cfg)
    assert torch.equal(y, ref) and torch.equal(ys.view(torch.int32), ref_s.view(torch.int32))
    with pytest.raises(Exception):
        ops.conv2d_split(x, w, None, None, 1, None, 0, 1, cfg, give_fp32=False)


@pytest.mark.parametrize("cin,cout,H,W", [(196, 256, 12, 16), (128, 196, 20, 12)])
def test_conv1x1_bilinear_residual(cin, cout, H, W):
    """lateral 1x1 conv + align_corners=True x2 upsample of the coarser map (resnet.py:151-157)."""
    from tests import hip_ops as ops
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(1, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 1, 1, generator=g) * (1.0 / cin) ** 0.5
    low = torch.randn(1, cout, H // 2, W // 2, generator=g)
    ref = F.conv2d(x, w) + F.interpolate(low, scale_factor=2.0, mode="bilinear", align_corners=True)
    got, pad_max = ops.conv2d(x, w, None, None, 1, low, 2, 0, -1)
    assert pad_max == 0.0
    assert (got - ref).abs().max() <= 2e-5 * max(1.0, ref.abs().max().item())


def test_conv_leaky():
    from tests import hip_ops as ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 128, 9, 11, generator=g)
    w = torch.randn(128, 128, 3, 3, generator=g) * 0.03
    ref = F.leaky_relu(F.conv2d(x.double(), w.double(), None, 1, 1), 0.01).float()
    got, _ = ops.conv2d(x, w, None, None, 1, None, 0, 2, -1)
    assert (got - ref).abs().max() <= 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("C", [128, 256])
def test_layernorm(C):
    from tests import hip_ops as ops
    g = torch.Generator().manual_seed(C)
    x = torch.randn(517, C, generator=g) * 3 + 1
    gam, bet = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    res = torch.randn(517, C, generator=g)
    ref = F.layer_norm(x.double(), (C,), gam.double(), bet.double(), 1e-5)
    assert (ops.layer_norm(x, gam, bet) - ref.float()).abs().max() < 2e-5
    assert (ops.layer_norm(x, gam, bet, res) - (res.double() + ref).float()).abs().max() < 2e-5


@pytest.mark.parametrize("M,K,N", [(300, 256, 256), (9096, 512, 256), (64, 256, 256), (77, 128, 128), (1000, 256, 128)])
@pytest.mark.parametrize("h2", [0, 1, 3])
@pytest.mark.parametrize("with_res", [False, True])
def test_linear_layernorm_fused(M, K, N, h2, with_res):
    """GEMM with the LayerNorm (+ residual) of LoFTREncoderLayer fused into its epilogue vs fp64."""
    from tests import hip_ops as ops
    if h2 == 1:
        _needs_fp16x2()
    g = torch.Generator().manual_seed(M + K + N + h2)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) * (1.0 / K) ** 0.5
    gamma = torch.rand(N, generator=g) + 0.5
    beta = torch.randn(N, generator=g) * 0.1
    res = torch.randn(M, N, generator=g) if with_res else None
    y = A.double() @ W.double().T
    ref = F.layer_norm(y, (N,), gamma.double(), beta.double(), 1e-5)
    if with_res:
        ref = res.double() + ref
    got = ops.linear_layernorm(A, W, gamma, beta, res, h2=h2, in_place=with_res)
    assert torch.isfinite(got).all()
    assert (got.double() - ref).abs().max() < 2e-5, float((got.double() - ref).abs().max())


@pytest.mark.parametrize("mag", [1e-5, 0.02, 1.0, 700.0])
@pytest.mark.parametrize("scaled", [True, False])
def test_pack_h2_bit_exact_vs_oracle(mag, scaled):
    """`opp_pack_h2` (weight pre-split for the fp16x2 GEMMs): byte-exact against oracle/fp16x2_oracle.py."""
    import numpy as np
    from oracle import fp16x2_oracle as X
    from tests import hip_ops as ops
    g = torch.Generator().manual_seed(int(mag * 1000) + 7)
    w = torch.randn(192, 1152, generator=g) * mag
    w[3, 5] = 0.0
    w[7, :8] = torch.tensor([mag * 2.0 ** -12, -mag * 2.0 ** -20, 1e-30, -1e-30, mag, -mag, mag / 3, -mag / 7])
    out, sc = ops.pack_h2(w.cuda(), scaled=scaled)
    torch.cuda.synchronize()
    img, s, inv = X.split_weights(w.numpy(), scaled=scaled)
    got = out.cpu().numpy().view(np.uint16).reshape(-1, 16)
    assert np.array_equal(got, img)
    if scaled:
        assert sc.cpu().tolist() == [float(s), float(inv)]


@pytest.mark.parametrize("h2", [3, 0])
def test_results_do_not_depend_on_the_tile_shape(h2):
    """Every tile configuration walks K in the same order with the same product sequence and the epilogues are
    compiled without FMA contraction where it matters (bilinear residual): outputs are bit-identical across tiles.
    That is what lets `opp_config.tile_policy` (latency / throughput tile choice) leave results untouched and what
    makes exact confidence ties behave like the reference's."""
    from tests import hip_ops as ops
    g = torch.Generator().manual_seed(1)
    cfgs = (25, 26, 22, 20, 2) if h2 != 0 else (25, 26, 20, 22, 2, 1, 0)
    for (M, K, N) in [(1000, 512, 512), (156, 128, 384), (4096, 64, 128), (700, 256, 196)]:
        A, W = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
        dcfgs = cfgs + ((27,) if (h2 == 3 and N <= 208) else ())          # 128 x 224 ring tile: <= 208 real columns
        outs = [ops.linear(A, W, 1, c, h2=h2) for c in dcfgs]
        assert all(torch.equal(outs[0], o) for o in outs[1:]), (M, K, N)
    for (cin, cout, ks, stride, H, Wd) in [(128, 128, 3, 1, 64, 64), (128, 196, 3, 2, 64, 64), (196, 196, 3, 1, 32, 48), (196, 256, 1, 1, 16, 24)]:
        x = torch.randn(1, cin, H, Wd, generator=g)
        w = torch.randn(cout, cin, ks, ks, generator=g) * 0.03
        scale, bias = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
        Ho, Wo = (H + 2 * (ks // 2) - ks) // stride + 1, (Wd + 2 * (ks // 2) - ks) // stride + 1
        res = torch.randn(1, cout, Ho, Wo, generator=g)
        ccfgs = cfgs + ((27,) if (h2 == 3 and cout == 196) else ())      # 128 x 224 ring tile: the 196(->224)-column layers only
        outs = [ops.conv2d(x, w, scale, bias, stride, res, 1, 1, c, h2=h2)[0] for c in ccfgs]
        assert all(torch.equal(outs[0], o) for o in outs[1:]), (cin, cout, ks, stride)
        if ks == 1 and stride == 1:      # FPN lateral: 1x1 conv + bilinear x2 residual
            low = torch.randn(1, cout, Ho // 2, Wo // 2, generator=g)
            outs = [ops.conv2d(x, w, None, None, 1, low, 2, 0, c, h2=h2)[0] for c in cfgs]
            assert all(torch.equal(outs[0], o) for o in outs[1:]), "bilinear residual"


# ---- LinearAttention on its own (loftr_module/linear_attention.py:29-61) --------------------------------------------
LINATTN_CASES = [
    # n_seg, len0, len1, C, nhead            code path
    (1, 4096, 5000, 256, 8),               # coarse level: MFMA gather + MFMA apply, both streams per launch
    (1, 300, 77, 256, 8),                  # ragged chunk tails
    (1, 33, 1, 256, 8),
    (37, 25, 1, 128, 8),                   # fine level: one workgroup per match
    (5, 9, 23, 128, 8),
    (3, 100, 40, 128, 8),                  # generic path (chunk partials + fixed-order reduction)
    (2, 640, 300, 256, 8),                 # B > 1 at the coarse level
]


def _linattn_reference(qkv, n_seg, len0, len1, C, nhead, cross):
    """fp64 restatement of LinearAttention.forward on already-activated inputs: Q = phi(q), K = phi(k), values = v / S;
    message = Q (K^T values) / (Q . sum K + eps) * S_src per head."""
    D = C // nhead
    q = qkv.double()
    T0 = n_seg * len0
    streams = [q[:T0].view(n_seg, len0, 3, nhead, D), q[T0:].view(n_seg, len1, 3, nhead, D)]
    out = []
    for st in (0, 1):
        src = streams[1 - st] if cross else streams[st]
        Q, K, V = streams[st][:, :, 0], src[:, :, 1], src[:, :, 2]
        KV = torch.einsum("nshd,nshv->nhdv", K, V)
        Z = 1.0 / (torch.einsum("nlhd,nhd->nlh", Q, K.sum(1)) + 1e-6)
        out.append((torch.einsum("nlhd,nhdv,nlh->nlhv", Q, KV, Z) * src.shape[1]).reshape(-1, C))
    return torch.cat(out, 0)


@pytest.mark.parametrize("n_seg,len0,len1,C,nhead", LINATTN_CASES)
@pytest.mark.parametrize("cross", [0, 1])
def test_linear_attention_vs_fp64(n_seg, len0, len1, C, nhead, cross):
    from tests import hip_ops as ops
    g = torch.Generator().manual_seed(n_seg * 131 + len0 + 7 * len1 + cross)
    T0, T1 = n_seg * len0, n_seg * len1
    raw = torch.randn(T0 + T1, 3 * C, generator=g)
    qkv = raw.clone()
    qkv[:, :2 * C] = F.elu(raw[:, :2 * C]) + 1.0                      # phi (linear_attention.py:5-6)
    qkv[:T0, 2 * C:] = raw[:T0, 2 * C:] / len0                        # values / S (linear_attention.py:55-56)
    qkv[T0:, 2 * C:] = raw[T0:, 2 * C:] / len1
    ref = _linattn_reference(qkv, n_seg, len0, len1, C, nhead, cross)
    got = ops.linear_attention(qkv, n_seg, len0, len1, C, nhead, cross)
    assert torch.isfinite(got).all()
    err = (got.double() - ref).abs().max().item()
    assert err <= 2e-5 * max(1.0, ref.abs().max().item()), err
    # zeroed K / V rows (what query_image_mask does to masked image tokens) contribute nothing
    if len0 >= 8:
        masked = qkv.clone()
        drop = torch.arange(0, len0, 3)
        masked.view(-1, 3 * C)[drop, C:] = 0.0
        dense = masked[:T0].view(n_seg, len0, 3 * C)
        keep = torch.ones(len0, dtype=torch.bool)
        keep[drop] = False
        if n_seg == 1 and cross:   # stream 1 attends to the kept rows of stream 0 only
            got_m = ops.linear_attention(masked, n_seg, len0, len1, C, nhead, cross)
            sub = torch.cat([dense[:, keep].reshape(-1, 3 * C), masked[T0:]], 0)
            ref_m = _linattn_reference(sub, n_seg, int(keep.sum()), len1, C, nhead, cross)[-T1:]
            # the source length factor stays len0 (the reference multiplies by the padded length, linear_attention.py:59)
            ref_m = ref_m * (len0 / float(keep.sum()))
            assert (got_m[-T1:].double() - ref_m).abs().max().item() <= 2e-5 * max(1.0, ref_m.abs().max().item())


@pytest.mark.gpu
@pytest.mark.parametrize("which,n_seg,len0,len1", [(0, 1, 4096, 5000), (0, 1, 96, 77), (0, 1, 31, 1), (0, 1, 64, 33), (0, 1, 65, 129),
                                                    (1, 500, 25, 1), (1, 1, 25, 1), (1, 37, 25, 1)])
def test_encoder_chain_is_bit_identical_to_the_launch_per_linear_path(which, n_seg, len0, len1):
    """One LoFTREncoderLayer behind its Q/K/V projection as ONE kernel (enc_chain.hip: attention apply, merge, norm1,
    mlp.0, ReLU, mlp.2, norm2, residual on 32-token tiles held in LDS; transformer.py:80-94) against the same layers run
    Linear by Linear through opp_gemm_kernel: both walk K in the same k16-steps with the same six bf16 products, so the
    whole transformer (coarse: 6 layers on L + N tokens; fine: 2 layers on M x (25 + 1) tokens) must agree bit for bit,
    ragged token counts and tiles that end inside a 32-row block included."""
    from tests import hip_ops as ops
    from onepose_plus_plus_amd import default_config
    from onepose_plus_plus_amd.synthetic import make_state_dict
    cfg = default_config()
    sd = make_state_dict(cfg, 3)
    C = 256 if which == 0 else 128
    g = torch.Generator().manual_seed(11 + len0 + n_seg)
    tokens = torch.randn(n_seg * (len0 + len1), C, generator=g)
    plain = ops.make_model(cfg, sd, "bf16x3").set_encoder_fusion(0).cuda()
    b = ops.transformer(plain, which, tokens, n_seg, len0, len1)
    for level in (2, 1):       # 64-token tiles (coarse level; enc_layer64.hip) and 32-token tiles (enc_chain.hip)
        fused = ops.make_model(cfg, sd, "bf16x3").set_encoder_fusion(level).cuda()
        a = ops.transformer(fused, which, tokens, n_seg, len0, len1)
        assert torch.isfinite(a).all()
        assert not torch.equal(a, tokens)
        assert torch.equal(a, b), "level %d: max |fused - plain| = %.3e" % (level, (a - b).abs().max().item())


@pytest.mark.gpu
@pytest.mark.parametrize("prec", [2, 0])
@pytest.mark.parametrize("M,K,N", [(9096, 256, 768), (300, 128, 256), (44384, 512, 512), (33, 256, 256), (1, 128, 128), (4097, 512, 256)])
def test_linear_backward_vs_fp64(M, K, N, prec):
    """Backward of a bias-free Linear on the MFMA GEMM (csrc/linear_bwd.hip: grad_x = grad_y W, grad_W = grad_y^T x as a
    split-K reduction over the tokens with a fixed-order sum) through the autograd node the training step uses
    (train_autograd.HipLinear; loftr_module/transformer.py:26-47), against an fp64 evaluation: error relative to
    sum |a||b| below 1e-6 (fp32 accumulation over up to 44384 terms) in both arithmetics; two runs are bit-identical (deterministic reduction)."""
    from onepose_plus_plus_amd.train_autograd import HipLinear
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g).cuda().requires_grad_(True)
    w = (torch.randn(N, K, generator=g) * 0.05).cuda().requires_grad_(True)
    gy = torch.randn(M, N, generator=g).cuda()
    y = HipLinear.apply(x, w, prec)
    y.backward(gy)
    torch.cuda.synchronize()
    xd, wd, gd = x.detach().double(), w.detach().double(), gy.double()
    ref_y, ref_dx, ref_dw = xd @ wd.T, gd @ wd, gd.T @ xd
    sy, sx, sw = xd.abs() @ wd.abs().T, gd.abs() @ wd.abs(), gd.abs().T @ xd.abs()
    assert ((y.detach().double() - ref_y).abs() / sy).max() < 1e-6
    assert ((x.grad.double() - ref_dx).abs() / sx).max() < 1e-6
    assert ((w.grad.double() - ref_dw).abs() / sw).max() < 1e-6, ((w.grad.double() - ref_dw).abs() / sw).max()
    x2, w2 = x.detach().clone().requires_grad_(True), w.detach().clone().requires_grad_(True)
    HipLinear.apply(x2, w2, prec).backward(gy)
    torch.cuda.synchronize()
    assert torch.equal(x2.grad, x.grad) and torch.equal(w2.grad, w.grad)


@pytest.mark.gpu
@pytest.mark.parametrize("B,L,S,H,D,masked", [(2, 96, 77, 8, 32, False), (1, 4096, 5000, 8, 32, True), (3, 25, 1, 8, 16, False),
                                               (4, 1, 25, 8, 16, False), (2, 130, 64, 8, 32, True),
                                               # short segments (L, S <= 32): one workgroup per sample, one launch per pass (la_small_kernel)
                                               (5, 25, 25, 8, 16, True), (2, 7, 30, 8, 32, True), (1500, 25, 1, 8, 16, False), (3, 32, 32, 4, 32, False)])
def test_linear_attention_backward_vs_fp64(B, L, S, H, D, masked):
    """LinearAttention.forward with its backward in libopp_hip.so (csrc/linattn_train.hip; the node the training step's graph uses
    instead of three einsums, loftr_module/linear_attention.py:29-61) against torch autograd on an fp64 evaluation of the
    reference formula: output and the three input gradients, with and without masks, coarse (D = 32) and fine (D = 16) heads;
    two runs are bit-identical (chunk partials summed in order)."""
    from onepose_plus_plus_amd.train_autograd import HipLinearAttention, _linear_attention
    g = torch.Generator().manual_seed(B * 1000 + L + S)
    q = torch.randn(B, L, H, D, generator=g).cuda().requires_grad_(True)
    k = torch.randn(B, S, H, D, generator=g).cuda().requires_grad_(True)
    v = torch.randn(B, S, H, D, generator=g).cuda().requires_grad_(True)
    go = torch.randn(B, L, H, D, generator=g).cuda()
    qm = (torch.rand(B, L, generator=g) < 0.8).float().cuda() if masked else None
    km = (torch.rand(B, S, generator=g) < 0.8).float().cuda() if masked else None
    out = HipLinearAttention.apply(q, k, v, qm, km)
    out.backward(go)
    torch.cuda.synchronize()
    qd, kd, vd = (t.detach().double().requires_grad_(True) for t in (q, k, v))
    ref = _linear_attention(qd, kd, vd, qm.double() if masked else None, km.double() if masked else None)     # fp64 tensors: the torch formula
    ref.backward(go.double())
    for got, want, what in ((out.detach(), ref.detach(), "out"), (q.grad, qd.grad, "gq"), (k.grad, kd.grad, "gk"), (v.grad, vd.grad, "gv")):
        err = (got.double() - want).abs().max().item()
        assert err <= 2e-5 * max(want.abs().max().item(), 1.0), (what, err, want.abs().max().item())     # O(1) inputs: fp32-level absolute floor
    q2, k2, v2 = (t.detach().clone().requires_grad_(True) for t in (q, k, v))
    HipLinearAttention.apply(q2, k2, v2, qm, km).backward(go)
    torch.cuda.synchronize()
    assert torch.equal(q2.grad, q.grad) and torch.equal(k2.grad, k.grad) and torch.equal(v2.grad, v.grad)


@pytest.mark.gpu
@pytest.mark.parametrize("M,K,N", [(50, 32, 7), (129, 96, 208), (1000, 1152, 130), (64, 160, 196), (300, 64, 224)])
def test_ring_tile_edge_shapes_dense(M, K, N):
    """128 x 224 ring tile (config 27) outside the shapes the model gives it: one chunk, odd chunk counts, partial row tiles, the 208-column limit.
    Bit-identical to the 128 x 256 tile; more than 208 real columns are refused (rows >= 208 are never loaded)."""
    from tests import hip_ops as ops
    g = torch.Generator().manual_seed(7 * M + K + N)
    A, W = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    if N > 208:
        W[208:] = 0.0                                  # the caller of an explicit 27 vouches for zero rows >= 208 ...
        got = ops.linear(A, W, 1, 27, h2=3)            # ... and then gets the 128 x 256 result
        assert torch.equal(got, ops.linear(A, W, 1, 22, h2=3))
        return
    assert torch.equal(ops.linear(A, W, 1, 27, h2=3), ops.linear(A, W, 1, 22, h2=3))


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,ks,stride,H,Wd", [(32, 196, 1, 1, 8, 8), (196, 196, 3, 2, 17, 23), (256, 196, 3, 1, 10, 12), (128, 200, 3, 1, 9, 9)])
def test_ring_tile_edge_shapes_conv(cin, cout, ks, stride, H, Wd):
    from tests import hip_ops as ops
    g = torch.Generator().manual_seed(cin + cout + H)
    x = torch.randn(1, cin, H, Wd, generator=g)
    w = torch.randn(cout, cin, ks, ks, generator=g) * 0.05
    scale, bias = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    a, pa = ops.conv2d(x, w, scale, bias, stride, None, 0, 1, 27, h2=3)
    b, pb = ops.conv2d(x, w, scale, bias, stride, None, 0, 1, 25, h2=3)
    assert torch.equal(a, b) and pa == 0.0 and pb == 0.0
