#!/usr/bin/env python3
"""GPU probe of the strict engine (k_trunk_split_c128): trunk launch time at the bench's batch against the 16-bit kernel
(interleaved, HIP events), and the error of every engine on trained-like weights against the fp32 restatement.

    python tools/strict_probe.py [B=8192] [iters=10]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import nethelpers as H  # noqa: E402
from cchess_zero_amd.net import PolicyValueNet, flops_per_position  # noqa: E402
from oracle import net_numpy  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    x = H.positions(64, 2)
    for blocks in (7, 19):
        for dname, dt, split in (("fp16", torch.float16, False), ("fp16x2", torch.float16, True), ("bf16x2", torch.bfloat16, True)):
            for wset in ("glorot", "trained_like"):
                net = PolicyValueNet(blocks, "cuda:0", dt, seed=1, split=split)
                H.WEIGHT_SETS[wset](net)
                logits, v = net.forward(x)
                ln, vn = net_numpy.forward(net.module.export_tf_layout(), x, blocks)
                e = H.errors(logits, v, ln, vn)
                print("%-7s %2d-block %-12s max|logit| %7.3g dlogit %.3g dvalue %.3g dprob %.3g argmax %.3f" %
                      (dname, blocks, wset, e["max_abs_logit"], e["dlogit"], e["dvalue"], e["dprob"], e["argmax_agree"]), flush=True)
    # timing: real planes of corpus positions tiled to B rows, trained-like weights (finite, peaked activations)
    xs = torch.from_numpy(np.tile(H.positions(256, 3), (B // 256 + 1, 1, 1, 1))[:B]).cuda()
    for blocks in (7, 19):
        nets = {}
        for dname, dt, split in (("fp16", torch.float16, False), ("fp16x2", torch.float16, True), ("bf16x2", torch.bfloat16, True)):
            n = PolicyValueNet(blocks, "cuda:0", dt, seed=1, split=split)
            H.trained_like_(n)
            p16 = torch.zeros((B, 9, 10, 16), dtype=dt, device="cuda")
            p16[..., :14] = xs.to(dt)
            nets[dname] = (n, p16)
        for rep in range(3):
            for dname, (n, p16) in nets.items():
                for _ in range(2):
                    n._hip_net_forward(p16)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    n._hip_net_forward(p16)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / iters
                fl = (flops_per_position(blocks) - 2 * (180 * 2086 + 90 * 256 + 256)) * B
                print("trunk %-7s %2d blocks B=%d: %8.1f us per launch = %6.0f algorithmic TFLOP/s, %.3f M positions/s" %
                      (dname, blocks, B, us, fl / us / 1e6, B / us), flush=True)


if __name__ == "__main__":
    main()
