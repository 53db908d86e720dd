#!/usr/bin/env python3
"""CPU emulation of what 16-bit storage of the tower activations does to a SEARCH (no GPU needed): the C oracle search
driven by the fp32 torch module vs the same module with weights and per-layer activations rounded to bf16 / fp16
(fp32 accumulate — the rounding points of the fused MFMA kernel), trained-like weights (tests/nethelpers.py).
Prints the root-argmax agreement and the L1 distance of the visit distributions: where the thresholds of
tests/test_bench_path.py::test_fused_search_agrees_with_fp32_engine come from.

    python tests/agree_emulation.py [bf16|fp16] [trees] [playouts]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # tests/ -> repo root
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import nethelpers as H  # noqa: E402
from cchess_zero_amd.net import PolicyValueModule  # noqa: E402
from oracle import oracle as O  # noqa: E402


def rounded_forward(m, x, dt):
    """PolicyValueModule.forward with BN folded, conv weights and the activation after every ReLU rounded to dt."""
    r = lambda t: t.to(dt).float()

    def cb(c, h, res=None):
        w, b = c.folded()
        y = F.conv2d(h, r(w), b, padding=w.shape[-1] // 2)
        return y if res is None else y + res
    h = r(torch.relu(cb(m.conv_in, r(x))))
    for a, b in m.blocks:
        t = r(torch.relu(cb(a, h)))
        h = r(torch.relu(cb(b, t, res=h)))
    wp, bp = m.policy_conv.folded()
    wv, bv = m.value_conv.folded()
    p = torch.relu(F.conv2d(h, wp, bp)).permute(0, 2, 3, 1).reshape(h.shape[0], 180)
    v = torch.relu(F.conv2d(h, wv, bv)).permute(0, 2, 3, 1).reshape(h.shape[0], 90)
    return m.policy_fc(p), torch.tanh(m.value_fc2(torch.relu(m.value_fc1(v))))


def main():
    dt = torch.float16 if (len(sys.argv) > 1 and sys.argv[1] == "fp16") else torch.bfloat16
    G = int(sys.argv[2]) if len(sys.argv) > 2 else 48
    playouts = int(sys.argv[3]) if len(sys.argv) > 3 else 400
    g = np.load(os.path.join(ROOT, "tests", "golden", "rules.npz"))
    ok = [(g["boards"][i] == 1).any() and (g["boards"][i] == 8).any() and g["counts"][i] > 0 for i in range(len(g["boards"]))]
    idx = np.nonzero(ok)[0][::37][:G]
    boards, side, rr = g["boards"][idx], g["side"][idx], np.zeros(len(idx), np.int32)

    class _N:
        pass
    n = _N()
    n.module = PolicyValueModule(7, seed=3)
    n.refresh = lambda: None
    H.trained_like_(n)
    res = {}
    for name in ("f32", "lo"):
        s = O.Search(len(idx), (playouts + 2) * 80)
        s.reset(boards, side, rr)
        for step in range(playouts + 1):
            planes, _ = s.select(0 if step == 0 else 1)
            xt = torch.from_numpy(planes).permute(0, 3, 1, 2)
            with torch.no_grad():
                l, v = n.module(xt) if name == "f32" else rounded_forward(n.module, xt, dt)
            s.expand_backup(l.numpy(), v.numpy())
        res[name] = s.root_stats()
    Na, Nb = res["lo"]["N"].astype(np.int64), res["f32"]["N"].astype(np.int64)
    l1 = np.abs(Na - Nb).sum(1) / playouts
    print("%s vs fp32, %d trees x %d playouts: argmax agreement %.4f, visit L1 mean %.4f max %.4f, top-move share %.3f" %
          (dt, len(idx), playouts, (Na.argmax(1) == Nb.argmax(1)).mean(), l1.mean(), l1.max(), (np.sort(Nb, 1)[:, -1] / playouts).mean()))


if __name__ == "__main__":
    main()
