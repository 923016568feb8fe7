"""GPU-side test wrappers around the C ABI (include/opp_hip.h).  Test infrastructure only."""
import ctypes

import torch

from onepose_plus_plus_amd import _lib, OnePosePlus_model


def _s():
    return torch.cuda.current_stream().cuda_stream


def pad32(c):
    return (c + 31) // 32 * 32


def pack_h2(w, scaled=True):
    """-> (pre-split weights, device {scale, 1/scale} or None)"""
    lib = _lib.load()
    out = torch.empty_like(w)
    sc = torch.zeros(2, device="cuda") if scaled else None
    _lib.check(lib.opp_pack_h2(w.data_ptr(), out.data_ptr(), w.numel(), sc.data_ptr() if scaled else None, _s()),
               "opp_pack_h2")
    return out, sc


def pack_b3(w):
    """-> bf16x3 pre-split weights (1.5x the floats)"""
    lib = _lib.load()
    out = torch.empty(w.numel() // 2 * 3, device="cuda")
    _lib.check(lib.opp_pack_b3(w.data_ptr(), out.data_ptr(), w.numel(), _s()), "opp_pack_b3")
    return out


def _prec(h2):
    """test-side selector -> C ABI `prec`: 0 fp32, 1 fp16x2 (scaled split), 2 fp16x2 (unscaled split), 3 bf16x3"""
    return {0: 0, 1: 1, 2: 1, 3: 2}[h2]


def _split(W, h2):
    if h2 == 3:
        return pack_b3(W), None
    if h2:
        return pack_h2(W, scaled=(h2 == 1))
    return W, None


def linear(A, W, act=0, cfg=-1, h2=0):
    lib = _lib.load()
    A = A.cuda().contiguous()
    W = W.cuda().contiguous()
    M, K = A.shape
    N = W.shape[0]
    W, sc = _split(W, h2)
    C = torch.full((M, N), float("nan"), device="cuda")
    _lib.check(lib.opp_linear(A.data_ptr(), M, K, W.data_ptr(), N, act, C.data_ptr(), cfg, _prec(h2),
                              sc.data_ptr() if sc is not None else None, _s()), "opp_linear")
    torch.cuda.synchronize()
    return C.cpu()


def linear_layernorm(A, W, gamma, beta, residual=None, h2=0, in_place=False):
    lib = _lib.load()
    A = A.cuda().contiguous()
    W = W.cuda().contiguous()
    M, K = A.shape
    N = W.shape[0]
    W, sc = _split(W, h2)
    g, b = gamma.cuda().contiguous(), beta.cuda().contiguous()
    r = residual.cuda().contiguous() if residual is not None else None
    C = r if (in_place and r is not None) else torch.full((M, N), float("nan"), device="cuda")
    _lib.check(lib.opp_linear_layernorm(A.data_ptr(), M, K, W.data_ptr(), N, g.data_ptr(), b.data_ptr(),
                                        r.data_ptr() if r is not None else None, C.data_ptr(), _prec(h2),
                                        sc.data_ptr() if sc is not None else None, _s()), "opp_linear_layernorm")
    torch.cuda.synchronize()
    return C.cpu()


def to_nhwc_padded(x_nchw, c_pad):
    b, c, h, w = x_nchw.shape
    assert b == 1
    out = torch.zeros(h, w, c_pad)
    out[:, :, :c] = x_nchw[0].permute(1, 2, 0)
    return out.cuda().contiguous()


def from_nhwc(y, c):
    return y[:, :, :c].permute(2, 0, 1).unsqueeze(0).cpu()


def conv2d(x_nchw, w, scale=None, bias=None, stride=1, residual=None, res_mode=0, act=0, cfg=-1, h2=0):
    """x [1,Cin,H,W]; w [Cout,Cin,k,k]; y = act(conv(x, w*scale) + bias + residual)."""
    lib = _lib.load()
    cout, cin, ks, _ = w.shape
    cin_p, cout_p = pad32(cin), pad32(cout)
    x = to_nhwc_padded(x_nchw, cin_p)
    H, W = x_nchw.shape[2:]
    pad = ks // 2
    Ho, Wo = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
    wp = torch.empty(cout_p * lib.opp_conv_packed_k(cin, ks), device="cuda")
    wd = w.cuda().contiguous()
    sd = None
    if scale is not None:
        sd = torch.zeros(cout_p, device="cuda")
        sd[:cout] = scale.cuda()
    _lib.check(lib.opp_pack_conv_weight(wd.data_ptr(), sd.data_ptr() if sd is not None else None, cout, cin, ks,
                                        cout_p, cin_p, wp.data_ptr(), _s()), "pack")
    wp, sc = _split(wp, h2)
    bd = None
    if bias is not None:
        bd = torch.zeros(cout_p, device="cuda")
        bd[:cout] = bias.cuda()
    rd = None
    if residual is not None:
        rd = to_nhwc_padded(residual, cout_p)
    y = torch.full((Ho, Wo, cout_p), float("nan"), device="cuda")
    _lib.check(lib.opp_conv2d_nhwc(x.data_ptr(), H, W, cin, wp.data_ptr(), bd.data_ptr() if bd is not None else None,
                                   cout_p, ks, stride, rd.data_ptr() if rd is not None else None, res_mode, act,
                                   y.data_ptr(), cfg, _prec(h2), sc.data_ptr() if sc is not None else None, _s()),
               "conv2d")
    torch.cuda.synchronize()
    pad_part = y[:, :, cout:]
    return from_nhwc(y, cout), (pad_part.abs().max().item() if pad_part.numel() else 0.0)


def conv2d_split(x_nchw, w, scale=None, bias=None, stride=1, residual=None, res_mode=0, act=0, cfg=-1, give_fp32=True, give_split=True,
                 want_fp32=True, want_split=True):
    """The bf16x3 convolution through opp_conv2d_nhwc_split: input as fp32 rows and / or pre-split rows (opp_pack_b3 of the fp32 rows), output
    as fp32 rows and / or pre-split rows.  -> (y fp32 NHWC padded or None, y_split raw floats or None, the plain opp_conv2d_nhwc(prec=2) result,
    opp_pack_b3 of that result)"""
    lib = _lib.load()
    cout, cin, ks, _ = w.shape
    cin_p, cout_p = pad32(cin), pad32(cout)
    x = to_nhwc_padded(x_nchw, cin_p)
    H, W = x_nchw.shape[2:]
    pad = ks // 2
    Ho, Wo = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
    wp = torch.empty(cout_p * lib.opp_conv_packed_k(cin, ks), device="cuda")
    wd = w.cuda().contiguous()
    sd = None
    if scale is not None:
        sd = torch.zeros(cout_p, device="cuda")
        sd[:cout] = scale.cuda()
    _lib.check(lib.opp_pack_conv_weight(wd.data_ptr(), sd.data_ptr() if sd is not None else None, cout, cin, ks,
                                        cout_p, cin_p, wp.data_ptr(), _s()), "pack")
    wp = pack_b3(wp)
    bd = None
    if bias is not None:
        bd = torch.zeros(cout_p, device="cuda")
        bd[:cout] = bias.cuda()
    rd = to_nhwc_padded(residual, cout_p) if residual is not None else None
    ref = torch.full((Ho, Wo, cout_p), float("nan"), device="cuda")
    _lib.check(lib.opp_conv2d_nhwc(x.data_ptr(), H, W, cin, wp.data_ptr(), bd.data_ptr() if bd is not None else None,
                                   cout_p, ks, stride, rd.data_ptr() if rd is not None else None, res_mode, act,
                                   ref.data_ptr(), cfg, 2, None, _s()), "conv2d")
    xs = pack_b3(x)
    y = torch.full((Ho, Wo, cout_p), float("nan"), device="cuda") if want_fp32 else None
    ys = torch.full((Ho * Wo * cout_p // 2 * 3,), float("nan"), device="cuda") if want_split else None
    _lib.check(lib.opp_conv2d_nhwc_split(x.data_ptr() if give_fp32 else None, xs.data_ptr() if give_split else None, H, W, cin, wp.data_ptr(),
                                         bd.data_ptr() if bd is not None else None, cout_p, ks, stride,
                                         rd.data_ptr() if rd is not None else None, res_mode, act,
                                         y.data_ptr() if y is not None else None, ys.data_ptr() if ys is not None else None, cfg, _s()),
               "conv2d_split")
    torch.cuda.synchronize()
    return y, ys, ref, pack_b3(ref)


def layer_norm(x, g, b, res=None):
    lib = _lib.load()
    x = x.cuda().contiguous()
    out = torch.empty_like(x)
    r = res.cuda().contiguous() if res is not None else None
    g, b = g.cuda().contiguous(), b.cuda().contiguous()
    _lib.check(lib.opp_layer_norm(x.data_ptr(), g.data_ptr(), b.data_ptr(), r.data_ptr() if r is not None else None,
                                  out.data_ptr(), x.shape[0], x.shape[1], _s()), "layer_norm")
    torch.cuda.synchronize()
    return out.cpu()


PRECISIONS = ("bf16x3", "fp32")        # the product's arithmetics (both not narrower than the reference's fp32)


def has_fp16x2():
    """fp16x2 kernels exist in the tuning library only (OPP_HIP_LIB=.../libopp_hip_tuning.so): kernel-level tests of that arithmetic skip otherwise"""
    return bool(_lib.load().opp_supports_precision(1))


def linear_attention(qkv, n_seg, len0, len1, C, nhead, cross):
    """qkv [n_seg*(len0+len1), 3C] (phi(Q) | phi(K) | V/S) -> message [same rows, C] (opp_linear_attention)."""
    lib = _lib.load()
    q = qkv.cuda().contiguous()
    msg = torch.full((q.shape[0], C), float("nan"), device="cuda")
    ws = torch.empty(lib.opp_linear_attention_workspace_bytes(n_seg, len0, len1, C, nhead), dtype=torch.uint8, device="cuda")
    _lib.check(lib.opp_linear_attention(q.data_ptr(), n_seg, len0, len1, C, nhead, cross, msg.data_ptr(), ws.data_ptr(),
                                        ws.numel(), _s()), "linear_attention")
    torch.cuda.synchronize()
    return msg.cpu()


def make_model(cfg, sd, precision=None):
    m = OnePosePlus_model(cfg).eval()
    m.load_state_dict(sd, strict=True)
    if precision is not None:
        m.set_gemm_precision(precision)
    return m.cuda()


def ctx_of(model):
    return model._ensure_ready(torch.device("cuda", torch.cuda.current_device()))


def backbone(model, img):
    """img [1,1,H,W] (cpu) -> feat_c NCHW [1,256,H/8,W/8], feat_f NCHW [1,128,H/2,W/2] (cpu)."""
    lib, ctx = ctx_of(model)
    H, W = img.shape[2:]
    x = img.cuda().contiguous()
    fc = torch.full((H // 8, W // 8, 256), float("nan"), device="cuda")
    ff = torch.full((H // 2, W // 2, 128), float("nan"), device="cuda")
    n = lib.opp_backbone_workspace_bytes(ctx, H, W)
    ws = torch.empty(n, dtype=torch.uint8, device="cuda")
    _lib.check(lib.opp_backbone(ctx, x.data_ptr(), H, W, fc.data_ptr(), ff.data_ptr(), ws.data_ptr(), n, _s()), "backbone")
    torch.cuda.synchronize()
    return from_nhwc(fc, 256), from_nhwc(ff, 128)


def coarse_tokens(model, feat_c_nchw, pe_tokens, kpts, bank_c):
    lib, ctx = ctx_of(model)
    hc, wc = feat_c_nchw.shape[2:]
    L = hc * wc
    N = kpts.shape[1]
    fc = to_nhwc_padded(feat_c_nchw, 256).reshape(L, 256)
    tok = torch.full((L + N, 256), float("nan"), device="cuda")
    ws = torch.empty(4096, dtype=torch.uint8, device="cuda")
    pe = pe_tokens.cuda().contiguous() if pe_tokens is not None else None
    k = kpts.cuda().contiguous()
    b = bank_c.cuda().contiguous()
    _lib.check(lib.opp_coarse_tokens(ctx, fc.data_ptr(), pe.data_ptr() if pe is not None else None, L, k.data_ptr(),
                                     b.data_ptr(), N, tok.data_ptr(), ws.data_ptr(), ws.numel(), _s()), "coarse_tokens")
    torch.cuda.synchronize()
    return tok.cpu()


def transformer(model, which, tokens, n_seg, len0, len1):
    lib, ctx = ctx_of(model)
    x = tokens.cuda().contiguous().clone()
    n = lib.opp_transformer_workspace_bytes(ctx, which, n_seg, len0, len1)
    ws = torch.empty(n, dtype=torch.uint8, device="cuda")
    _lib.check(lib.opp_transformer(ctx, which, x.data_ptr(), n_seg, len0, len1, ws.data_ptr(), n, _s()), "transformer")
    torch.cuda.synchronize()
    return x.cpu()


def coarse_match(model, f3d, f2d, hw_c, kpts, base_scale, qscale):
    """f3d [N,C], f2d [L,C] -> dict like the reference data updates."""
    lib, ctx = ctx_of(model)
    N, L = f3d.shape[0], f2d.shape[0]
    a, b, k = f3d.cuda().contiguous(), f2d.cuda().contiguous(), kpts.cuda().contiguous()
    q = qscale.cuda().contiguous() if qscale is not None else None
    conf = torch.full((1, N, L), float("nan"), device="cuda")
    i_ids = torch.empty(N, dtype=torch.int64, device="cuda")
    j_ids = torch.empty(N, dtype=torch.int64, device="cuda")
    mconf = torch.empty(N, device="cuda")
    mkc = torch.empty(N, 2, device="cuda")
    mk3 = torch.empty(N, 3, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    n = lib.opp_coarse_match_workspace_bytes(ctx, N, L)
    ws = torch.empty(n, dtype=torch.uint8, device="cuda")
    _lib.check(lib.opp_coarse_match(ctx, a.data_ptr(), b.data_ptr(), N, hw_c[0], hw_c[1], k.data_ptr(), base_scale,
                                    q.data_ptr() if q is not None else None, conf.data_ptr(), i_ids.data_ptr(),
                                    j_ids.data_ptr(), mconf.data_ptr(), mkc.data_ptr(), mk3.data_ptr(), cnt.data_ptr(),
                                    ws.data_ptr(), n, _s()), "coarse_match")
    torch.cuda.synchronize()
    M = int(cnt.item())
    bz = torch.zeros(M, dtype=torch.int64)
    return {"conf_matrix": conf.cpu(), "b_ids": bz, "i_ids": i_ids[:M].cpu(), "j_ids": j_ids[:M].cpu(),
            "mconf": mconf[:M].cpu(), "mkpts_query_c": mkc[:M].cpu(), "mkpts_3d_db": mk3[:M].cpu(),
            "m_bids": bz, "gt_mask": torch.zeros(M, dtype=torch.bool)}


def fine(model, feat_f_nchw, bank_f, i_ids, j_ids, hw_c, mkpts_c, base_scale, qscale, run_transformer=1):
    lib, ctx = ctx_of(model)
    hf, wf = feat_f_nchw.shape[2:]
    M = i_ids.numel()
    N = bank_f.shape[2]
    ff = to_nhwc_padded(feat_f_nchw, 128)
    bk = bank_f.cuda().contiguous()
    ii, jj = i_ids.cuda().contiguous(), j_ids.cuda().contiguous()
    mk = mkpts_c.cuda().float().contiguous()
    q = qscale.cuda().contiguous() if qscale is not None else None
    ex = torch.full((M, 3), float("nan"), device="cuda")
    mf = torch.full((M, 2), float("nan"), device="cuda")
    n = lib.opp_fine_workspace_bytes(ctx, M)
    ws = torch.empty(n, dtype=torch.uint8, device="cuda")
    _lib.check(lib.opp_fine(ctx, ff.data_ptr(), hf, wf, bk.data_ptr(), N, ii.data_ptr(), jj.data_ptr(), M, hw_c[0],
                            hw_c[1], mk.data_ptr(), base_scale, q.data_ptr() if q is not None else None,
                            run_transformer, ex.data_ptr(), mf.data_ptr(), ws.data_ptr(), n, _s()), "fine")
    torch.cuda.synchronize()
    return ex.cpu(), mf.cpu()


def run_model(model, data_cpu):
    d = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in data_cpu.items()}
    with torch.no_grad():
        model(d)
    torch.cuda.synchronize()
    return d
