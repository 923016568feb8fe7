"""Per-parameter error of the HIP backbone backward (HipBackbone) against fp64 autograd of the functional restatement, in network
order, for both arithmetics, plus run-to-run determinism.    python tools/bwd_diag.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from onepose_plus_plus_amd import train_autograd as TA  # noqa: E402
from tests import torch_graph_ref as GR  # noqa: E402
from onepose_plus_plus_amd.config import default_config  # noqa: E402
from onepose_plus_plus_amd.synthetic import make_state_dict  # noqa: E402
from tests import hip_ops as ops  # noqa: E402


def main():
    cfg = default_config()
    sd = make_state_dict(cfg, 4)
    B, H, W = 2, 64, 96
    g = torch.Generator().manual_seed(11)
    img = torch.rand(B, 1, H, W, generator=g)
    p = {k: v.double().requires_grad_(k.startswith("backbone.") and not k.endswith(("running_mean", "running_var", "num_batches_tracked")))
         for k, v in sd.items() if v.is_floating_point()}
    rc, rf = GR._backbone(p, img.double())
    rc_t, rf_t = rc.flatten(2).transpose(1, 2), rf.flatten(2).transpose(1, 2)
    gfc, gff = torch.randn(rc_t.shape, generator=g), torch.randn(rf_t.shape, generator=g)
    for which in ("both", "coarse_only", "fine_only"):
        for t in p.values():
            t.grad = None
        wc, wf = (1.0 if which != "fine_only" else 0.0), (1.0 if which != "coarse_only" else 0.0)
        ((rc_t * gfc.double()).sum() * wc + (rf_t * gff.double()).sum() * wf).backward(retain_graph=True)
        want = {k: v.grad.clone() for k, v in p.items() if v.grad is not None}
        for precision in ("bf16x3", "fp32"):
            model = ops.make_model(cfg, sd, precision)
            model.train()
            got = []
            for rep in range(2):
                for prm in model.parameters():
                    prm.grad = None
                lib, c = model._ensure_ready(torch.device("cuda:0"))
                fc, ff = TA.backbone_node(model, lib, c, img.cuda().contiguous())
                ((fc * gfc.cuda()).sum() * wc + (ff * gff.cuda()).sum() * wf).backward()
                torch.cuda.synchronize()
                got.append({n: q.grad.detach().cpu().double() for n, q in model.named_parameters() if q.grad is not None})
            print("== %s %s: forward rel err feat_c %.2e feat_f %.2e; two runs bit-identical: %s" % (
                which, precision, float((fc.detach().cpu().double() - rc_t.detach()).abs().max() / rc_t.detach().abs().max()),
                float((ff.detach().cpu().double() - rf_t.detach()).abs().max() / rf_t.detach().abs().max()),
                all(torch.equal(got[0][n], got[1][n]) for n in got[0])))
            for n in got[0]:
                w = want[n]
                e = float((got[0][n] - w).abs().max())
                print("   %-44s rel %.2e   (max |g| %.3e)" % (n, e / max(float(w.abs().max()), 1e-30), float(w.abs().max())))


if __name__ == "__main__":
    main()
