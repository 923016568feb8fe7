"""Stage-by-stage diagnostic on the GPU box (prints error statistics, never asserts)."""
import os
import sys
import time
import traceback

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests import helpers as H  # noqa: E402
from tests import hip_ops as ops  # noqa: E402
from oracle import onepose_oracle as O  # noqa: E402


def section(name, fn):
    print("=" * 20, name, flush=True)
    t = time.time()
    try:
        fn()
    except Exception:
        traceback.print_exc()
    print("   (%.2f s)" % (time.time() - t), flush=True)


def stat(name, got, ref):
    got, ref = got.float(), ref.float()
    d = (got - ref).abs()
    print("  %-28s max|err| %.3e  mean|err| %.3e  max|ref| %.3e  nan %d" %
          (name, d.max().item(), d.mean().item(), ref.abs().max().item(), int(torch.isnan(got).sum())), flush=True)


def t_linear():
    for cfg in (0, 1, 2, -1):
        g = torch.Generator().manual_seed(1)
        A, W = torch.randn(300, 256, generator=g), torch.randn(768, 256, generator=g)
        stat("linear cfg%d" % cfg, ops.linear(A, W, 0, cfg), (A.double() @ W.double().T).float())
    for cfg in (3, 5):
        g = torch.Generator().manual_seed(2)
        A, W = torch.randn(333, 96, generator=g), torch.randn(224, 96, generator=g)
        stat("linear224 cfg%d" % cfg, ops.linear(A, W, 0, cfg), (A.double() @ W.double().T).float())


def t_conv():
    for (cin, cout, ks, stride) in [(128, 128, 3, 1), (128, 196, 3, 2), (196, 256, 1, 1), (256, 256, 3, 1)]:
        g = torch.Generator().manual_seed(cin + cout)
        x = torch.randn(1, cin, 16, 24, generator=g)
        w = torch.randn(cout, cin, ks, ks, generator=g) * 0.05
        ref = F.conv2d(x.double(), w.double(), None, stride, ks // 2).float()
        got, padmax = ops.conv2d(x, w, None, None, stride, None, 0, 0, -1)
        stat("conv %d->%d k%d s%d (pad %.1e)" % (cin, cout, ks, stride, padmax), got, ref)


def t_stages():
    cfg, sd, data = H.e2e_setup("e2e_128x128_n300_thr0")
    model = ops.make_model(cfg, sd)
    with torch.no_grad():
        st = O.backbone_forward(sd, data["query_image"], stages=True)
    fc, ff = ops.backbone(model, data["query_image"])
    stat("backbone feat_c", fc, st["x3_out"])
    stat("backbone feat_f", ff, st["x1_out"])
    gold = H.load_golden("stages_128x128_n300")
    feat_c = torch.from_numpy(gold["feat_c"])
    hc, wc = feat_c.shape[2:]
    L = hc * wc
    pe = O.sine_position_table(256, (256, 256))[0, :, :hc, :wc].permute(1, 2, 0).reshape(L, 256)
    tok = ops.coarse_tokens(model, feat_c, pe, data["keypoints3d"], data["descriptors3d_coarse_db"])
    stat("tokens2d", tok[:L], torch.from_numpy(gold["tokens2d"][0]))
    ref3 = torch.from_numpy(gold["bank_enc"])[0].T
    stat("tokens3d", tok[L:], ref3)
    X = torch.cat([torch.from_numpy(gold["tokens2d"][0]), ref3], 0)
    out = ops.transformer(model, 0, X, 1, L, ref3.shape[0])
    stat("transformer f2", out[:L], torch.from_numpy(gold["f2"][0]))
    stat("transformer f3", out[L:], torch.from_numpy(gold["f3"][0]))
    for name in ("matcher_n700_p300", "matcher_n5000_p3000"):
        c2, f3d, f2d, d = H.matcher_setup(name)
        from onepose_plus_plus_amd.synthetic import make_state_dict
        mm = ops.make_model(c2, make_state_dict(c2, 0))
        got = ops.coarse_match(mm, f3d[0], f2d[0], tuple(d["q_hw_c"]), d["keypoints3d"][0], 8.0, d["query_image_scale"][0])
        g = H.load_golden(name)
        print("  %s: M got %d gold %d  ids equal %s" % (name, len(got["i_ids"]), len(g["i_ids"]),
              len(got["i_ids"]) == len(g["i_ids"]) and bool((got["i_ids"].numpy() == g["i_ids"]).all() and (got["j_ids"].numpy() == g["j_ids"]).all())))
        if len(got["mconf"]) == len(g["mconf"]):
            stat("  mconf", got["mconf"], torch.from_numpy(g["mconf"]))
        stat("  conf rowsum", got["conf_matrix"][0].sum(1), torch.from_numpy(g["conf_rowsum"]))
    for name in ("fine_m1", "fine_m500"):
        c2, sd2, feat_f, bank_f, d = H.fine_setup(name)
        ex, mf = ops.fine(model, feat_f, bank_f, d["i_ids"], d["j_ids"], tuple(d["q_hw_c"]), d["mkpts_query_c"], 2.0, d["query_image_scale"][0])
        g = H.load_golden(name)
        stat(name + " expec_f", ex, torch.from_numpy(g["expec_f"]))
        stat(name + " mkpts_f", mf, torch.from_numpy(g["mkpts_query_f"]))


def t_e2e():
    for name in ("e2e_128x128_n300_thr0", "e2e_64x96_n100_thr01", "e2e_512x512_n2000_thr0"):
        cfg, sd, data = H.e2e_setup(name)
        model = ops.make_model(cfg, sd)
        out = ops.run_model(model, data)
        g = H.load_golden(name)
        same = len(out["i_ids"]) == len(g["i_ids"]) and bool((out["i_ids"].cpu().numpy() == g["i_ids"]).all() and (out["j_ids"].cpu().numpy() == g["j_ids"]).all())
        print("  %s: M got %d gold %d ids equal %s" % (name, len(out["i_ids"]), len(g["i_ids"]), same))
        if same and len(g["mconf"]):
            stat("  mconf", out["mconf"].cpu(), torch.from_numpy(g["mconf"]))
            stat("  expec_f", out["expec_f"].cpu(), torch.from_numpy(g["expec_f"]))
            stat("  mkpts_f", out["mkpts_query_f"].cpu(), torch.from_numpy(g["mkpts_query_f"]))
        stat("  conf rowmax", out["conf_matrix"][0].max(1).values.cpu(), torch.from_numpy(g["conf_rowmax"]))


def t_timing():
    from onepose_plus_plus_amd.synthetic import make_state_dict, make_inputs
    from onepose_plus_plus_amd.config import default_config
    for fine in (False, True):
        cfg = default_config(thr=0.0, fine=fine)
        model = ops.make_model(cfg, make_state_dict(cfg, 0))
        d0 = {k: v.cuda() for k, v in make_inputs(5000, (512, 512), 1).items()}
        for _ in range(3):
            model(dict(d0))
        torch.cuda.synchronize()
        t = time.time()
        n = 20
        for _ in range(n):
            model(dict(d0))
        torch.cuda.synchronize()
        print("  512x512x5k fine=%s: %.3f ms / image" % (fine, (time.time() - t) / n * 1e3), flush=True)


if __name__ == "__main__":
    print(torch.__version__, torch.cuda.get_device_name(0), flush=True)
    section("linear", t_linear)
    section("conv", t_conv)
    section("stages", t_stages)
    section("e2e", t_e2e)
    section("timing", t_timing)
