"""fp32 NumPy restatement of the reference's TF1 policy/value graph (TEST INFRASTRUCTURE).

Follows policy_value_network.py:45-74 (graph), :151-162 (residual_block), :202-214 (forward) of
chengstone/cchess-zero.  The arithmetic lives in TensorFlow 1.x, which is not vendored in the
reference and not installable here, so this restatement is "parity unpinned" against real TF
outputs (stated in DESIGN.md); it pins the *graph semantics* the survey lists:
NHWC input [B,9,10,14], HWIO kernels, SAME padding, conv bias on, BN = (x-mean)/sqrt(var+1e-5)
without gamma/beta, flatten in (h,w,c) order before both FC stacks, un-softmaxed logits, tanh value.
Weights use the TF layout dict produced by PolicyValueModule.export_tf_layout().
"""
import numpy as np

EPS = np.float32(1e-5)


def conv2d_same(x, k, b):
    """x [B,H,W,Cin] f32, k [kh,kw,Cin,Cout] (HWIO), SAME padding, stride 1."""
    B, H, W, Cin = x.shape
    kh, kw, _, Cout = k.shape
    ph, pw = kh // 2, kw // 2
    xp = np.zeros((B, H + 2 * ph, W + 2 * pw, Cin), np.float32)
    xp[:, ph:ph + H, pw:pw + W] = x
    cols = np.empty((B, H, W, kh, kw, Cin), np.float32)
    for i in range(kh):
        for j in range(kw):
            cols[:, :, :, i, j, :] = xp[:, i:i + H, j:j + W, :]
    y = cols.reshape(B * H * W, kh * kw * Cin) @ k.reshape(kh * kw * Cin, Cout).astype(np.float32)
    return (y + b.astype(np.float32)).reshape(B, H, W, Cout)


def bn_inference(x, mean, var):
    return (x - mean.astype(np.float32)) / np.sqrt(var.astype(np.float32) + EPS)


def forward(weights, positions, res_block_nums):
    x = np.asarray(positions, np.float32)
    if x.ndim == 3:
        x = x[None]
    w = weights
    idx = [0]

    def convbn(h, relu):
        i = idx[0]
        idx[0] += 1
        y = conv2d_same(h, w["conv%d/kernel" % i], w["conv%d/bias" % i])
        y = bn_inference(y, w["bn%d/moving_mean" % i], w["bn%d/moving_variance" % i])
        return np.maximum(y, 0) if relu else y

    h = convbn(x, True)                      # policy_value_network.py:45-47
    for _ in range(res_block_nums):          # :151-162
        t = convbn(h, True)
        t = convbn(t, False)
        h = np.maximum(h + t, 0)
    p = convbn(h, True)                      # :57-59
    p = p.reshape(p.shape[0], 9 * 10 * 2)    # :62 (h,w,c order)
    logits = p @ w["policy_fc/weights"] + w["policy_fc/biases"]   # :63, no softmax
    v = convbn(h, True)                      # :68-70
    v = v.reshape(v.shape[0], 90)            # :72
    v = np.maximum(v @ w["value_fc1/weights"] + w["value_fc1/biases"], 0)   # :73
    v = np.tanh(v @ w["value_fc2/weights"] + w["value_fc2/biases"])          # :74
    return logits.astype(np.float32), v.astype(np.float32).reshape(-1, 1)
