"""fp16x2-split GEMM probe: accuracy vs fp64 at several operand magnitudes (subnormal behaviour of the
fp16 lo halves), and end-to-end agreement of the two precisions.   python tools/h2_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hip_ops  # noqa: E402
from onepose_plus_plus_amd import OnePosePlus_model  # noqa: E402
from onepose_plus_plus_amd.config import default_config  # noqa: E402
from onepose_plus_plus_amd.synthetic import make_inputs, make_state_dict  # noqa: E402


def main():
    torch.manual_seed(0)
    M, K, N = 512, 1152, 256
    for sa, sw in ((1.0, 1.0), (1.0, 0.02), (1e-2, 0.02), (1e-3, 0.02), (1e-4, 1.0), (30.0, 0.02), (1.0, 1e-4)):
        A = torch.randn(M, K) * sa
        W = torch.randn(N, K) * sw
        ref = A.double() @ W.double().t()
        norm = (A.double().abs() @ W.double().abs().t())
        for h2 in (0, 1):
            for cfg in (0, 1, 2, 11):
                C = hip_ops.linear(A, W, cfg=cfg, h2=h2).double()
                e = ((C - ref).abs() / norm).max().item()
                print("sa %-7g sw %-7g h2 %d cfg %2d  max |err| / sum|a||w| = %.3e" % (sa, sw, h2, cfg, e), flush=True)
    cfg = default_config(thr=0.0, fine=True)
    sd = make_state_dict(cfg, seed=3)
    outs = {}
    for prec in ("fp32", "fp16x2"):
        m = OnePosePlus_model(cfg)
        m.load_state_dict(sd)
        m = m.eval().cuda().set_gemm_precision(prec)
        data = make_inputs(5000, (512, 512), seed=5)
        data = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in data.items()}
        with torch.no_grad():
            m(data)
        outs[prec] = {k: data[k].detach().cpu() for k in ("i_ids", "j_ids", "mconf", "mkpts_query_f", "conf_matrix")
                      if k in data}
    a, b = outs["fp32"], outs["fp16x2"]
    print("matches fp32 %d fp16x2 %d" % (a["i_ids"].numel(), b["i_ids"].numel()))
    if a["i_ids"].shape == b["i_ids"].shape:
        print("ids equal:", bool((a["i_ids"] == b["i_ids"]).all() and (a["j_ids"] == b["j_ids"]).all()))
        print("max |d mconf| %.3e  max |d mkpts_query_f| %.3e" % ((a["mconf"] - b["mconf"]).abs().max().item(),
                                                               (a["mkpts_query_f"] - b["mkpts_query_f"]).abs().max().item()))
    if "conf_matrix" in a:
        print("max |d conf_matrix| %.3e" % (a["conf_matrix"] - b["conf_matrix"]).abs().max().item())


if __name__ == "__main__":
    main()
