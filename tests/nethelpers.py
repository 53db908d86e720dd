"""Shared by tests/test_net.py, tests/test_bench_path.py and tests/measure_net_errors.py: weight sets and error metrics
for the net parity checks (fused bf16 / fp16 MFMA net vs the fp32 NumPy restatement of the reference graph,
policy_value_network.py:45-74,151-162)."""
import os

import numpy as np
import torch


def positions(n, seed=0):
    """Real encoder outputs: planes of corpus positions (one-hot, with quirk Q1), [n,9,10,14] f32."""
    from oracle import oracle as O
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rules.npz"))
    rng = np.random.default_rng(seed)
    idx = rng.choice(len(g["boards"]), n, replace=False)
    return np.stack([O.encode_planes(g["boards"][i], int(g["side"][i])) for i in idx])


def structured_(net, seed=11):
    """Glorot weights + tap/channel-asymmetric perturbations, non-zero BN statistics and biases (catches tap,
    channel-order and residual mix-ups that symmetric noise could hide)."""
    gen = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for cb in net.module.convbns():
            dev = cb.conv.bias.device
            cb.moving_mean.copy_((torch.randn(cb.moving_mean.shape, generator=gen) * 0.05).to(dev))
            cb.moving_var.copy_((torch.rand(cb.moving_var.shape, generator=gen) + 0.5).to(dev))
            cb.conv.bias.copy_((torch.randn(cb.conv.bias.shape, generator=gen) * 0.05).to(dev))
            w = cb.conv.weight
            if w.shape[-1] == 3:
                w[:, :, 0, 1] += 0.01
                w[:, :, 2, 0] -= 0.015
                w[:, : w.shape[1] // 2, 1, 2] += 0.02
    net.refresh()
    return net


def trained_like_(net, seed=5):
    """Trained-like weight set (cchess_zero_amd.net.trained_like_), heads calibrated on 96 corpus positions."""
    from cchess_zero_amd.net import trained_like_ as tl
    return tl(net, positions(96, 123), seed)


WEIGHT_SETS = {"glorot": lambda net: net, "structured": structured_, "trained_like": trained_like_}


def softmax(a):
    e = np.exp(a - a.max(axis=1, keepdims=True))
    return e / e.sum(axis=1, keepdims=True)


def errors(logits, value, ref_logits, ref_value):
    """All the quantities the net tolerances are written in."""
    ml = float(np.abs(ref_logits).max())
    return dict(max_abs_logit=ml,
                dlogit=float(np.abs(logits - ref_logits).max()),
                dlogit_rel=float(np.abs(logits - ref_logits).max() / ml),
                dprob=float(np.abs(softmax(logits) - softmax(ref_logits)).max()),
                max_prob=float(softmax(ref_logits).max()),
                dvalue=float(np.abs(value - ref_value).max()),
                argmax_agree=float((logits.argmax(axis=1) == ref_logits.argmax(axis=1)).mean()))
