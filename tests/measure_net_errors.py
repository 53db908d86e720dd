#!/usr/bin/env python3
"""Measure, on the GPU, the error of every net engine against the fp32 NumPy restatement of the reference graph
(oracle/net_numpy.py) for the weight sets of tests/nethelpers.py; writes one JSON (default gpurun_out/net_errors.json).
The tolerance tables of tests/test_net.py are <= 3x these measured levels (profiles/r02_net_errors.json is the
committed copy they were derived from).

    python tests/measure_net_errors.py [out.json]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # tests/ -> repo root
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import nethelpers as H  # noqa: E402
from oracle import net_numpy  # noqa: E402


def main():
    from cchess_zero_amd.net import PolicyValueNet
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "net_errors.json")
    res = {}
    x = H.positions(64, 2)
    for dtype, dname, split in ((torch.bfloat16, "bf16", False), (torch.float16, "fp16", False), (torch.float32, "fp32", False),
                                (torch.float16, "fp16x2", True), (torch.bfloat16, "bf16x2", True)):
        for blocks in (2, 7, 19):
            for wname, wfn in H.WEIGHT_SETS.items():
                net = PolicyValueNet(blocks, "cuda:0", dtype, seed=1, split=split)
                wfn(net)
                logits, v = net.forward(x)
                ln, vn = net_numpy.forward(net.module.export_tf_layout(), x, blocks)
                e = H.errors(logits, v, ln, vn)
                e["backend"] = net.backend
                res["%s/%d/%s" % (dname, blocks, wname)] = e
                print("%-5s %2d-block %-12s max|logit| %8.3g dlogit %.3g (rel %.3g) dprob %.3g (max prob %.3g) dvalue %.3g argmax %.3f" %
                      (dname, blocks, wname, e["max_abs_logit"], e["dlogit"], e["dlogit_rel"], e["dprob"], e["max_prob"], e["dvalue"], e["argmax_agree"]), flush=True)
    # hip fused vs torch/MIOpen bf16 (same folded weights, different rounding points)
    for blocks in (2, 7, 19):
        a = PolicyValueNet(blocks, "cuda:0", torch.bfloat16, seed=1, backend="hip")
        b = PolicyValueNet(blocks, "cuda:0", torch.bfloat16, seed=1, backend="torch")
        xd = torch.from_numpy(x).cuda()
        la, va = a.forward_device(xd)
        lb, vb = b.forward_device(xd)
        e = H.errors(la.cpu().numpy(), va.cpu().numpy(), lb.cpu().numpy(), vb.cpu().numpy())
        ta, tb = a.tower(xd).float(), b.tower(xd).float()
        e["dtrunk_rel"] = float((ta - tb).abs().max() / tb.abs().max())
        res["hip_vs_torch_bf16/%d" % blocks] = e
        print("hip vs torch bf16 %2d-block: dlogit rel %.3g dprob %.3g dvalue %.3g dtrunk rel %.3g" % (blocks, e["dlogit_rel"], e["dprob"], e["dvalue"], e["dtrunk_rel"]), flush=True)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print("wrote", out)


if __name__ == "__main__":
    main()
