"""Residual policy/value network of cchess-zero re-expressed in PyTorch-ROCm.

Graph (policy_value_network.py:45-74,151-162 of the reference, TF1):
  conv3x3(14->128, bias) -> BN(no gamma/beta, eps 1e-5) -> ReLU
  N x [conv3x3+BN+ReLU, conv3x3+BN, add, ReLU]
  policy: conv1x1(128->2)+BN+ReLU -> flatten (h,w,c) 180 -> FC 2086 (raw logits, no softmax)
  value : conv1x1(128->1)+BN+ReLU -> flatten 90 -> FC 256 ReLU -> FC 1 tanh
Input is NHWC [B,9,10,14] (TF sees H=9, W=10; the board-to-plane indexing quirk Q1 lives in the
encoder, not here).  Inference folds BN into the conv (x - mean) / sqrt(var + eps); with the
reference's never-updated moving statistics (quirk Q5) that is x / sqrt(1 + 1e-5).

Weights are kept in torch layout (OIHW / [out,in]); export_tf_layout()/load_tf_layout() convert to
the reference's HWIO / [in,out] so a cchess-zero checkpoint's arrays can be dropped in.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

FILTERS = 128     # policy_value_network.py:23
PROB_SIZE = 2086  # policy_value_network.py:24
BN_EPS = 1e-5


def _glorot_(w, fan_in, fan_out, gen):
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    with torch.no_grad():
        w.copy_((torch.rand(w.shape, generator=gen, dtype=torch.float32) * 2 - 1) * lim)


class ConvBN(nn.Module):
    """tf.layers.conv2d(padding='SAME', bias) + tf.contrib.layers.batch_norm(center=False, scale=False)."""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, padding=k // 2, bias=True)
        self.register_buffer("moving_mean", torch.zeros(cout))
        self.register_buffer("moving_var", torch.ones(cout))

    def folded(self):
        s = torch.rsqrt(self.moving_var.float() + BN_EPS)
        w = self.conv.weight.float() * s.view(-1, 1, 1, 1)
        b = (self.conv.bias.float() - self.moving_mean.float()) * s
        return w, b

    def forward(self, x, training=False):
        y = self.conv(x)
        if training:  # TF is_training=True: batch statistics; the moving averages are never updated (Q5)
            m = y.mean(dim=(0, 2, 3), keepdim=True)
            v = y.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
            return (y - m) * torch.rsqrt(v + BN_EPS)
        return (y - self.moving_mean.view(1, -1, 1, 1)) * torch.rsqrt(self.moving_var.view(1, -1, 1, 1) + BN_EPS)


class PolicyValueModule(nn.Module):
    def __init__(self, res_block_nums=7, seed=0):
        super().__init__()
        self.res_block_nums = res_block_nums
        self.conv_in = ConvBN(14, FILTERS, 3)
        self.blocks = nn.ModuleList([nn.ModuleList([ConvBN(FILTERS, FILTERS, 3), ConvBN(FILTERS, FILTERS, 3)])
                                     for _ in range(res_block_nums)])
        self.policy_conv = ConvBN(FILTERS, 2, 1)
        self.policy_fc = nn.Linear(180, PROB_SIZE)
        self.value_conv = ConvBN(FILTERS, 1, 1)
        self.value_fc1 = nn.Linear(90, 256)
        self.value_fc2 = nn.Linear(256, 1)
        self.reset_parameters(seed)

    def reset_parameters(self, seed=0):
        """TF defaults: glorot_uniform kernels, zero biases."""
        gen = torch.Generator().manual_seed(seed)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                k = m.kernel_size[0] * m.kernel_size[1]
                _glorot_(m.weight, m.in_channels * k, m.out_channels * k, gen)
                nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Linear):
                _glorot_(m.weight, m.in_features, m.out_features, gen)
                nn.init.zeros_(m.bias)

    def convbns(self):
        out = [self.conv_in]
        for a, b in self.blocks:
            out += [a, b]
        return out + [self.policy_conv, self.value_conv]

    def forward(self, x_nchw, training=False):
        """Straight fp32/any-dtype module forward (used for training and as the torch reference)."""
        h = F.relu(self.conv_in(x_nchw, training))
        for a, b in self.blocks:
            t = F.relu(a(h, training))
            h = F.relu(h + b(t, training))
        p = F.relu(self.policy_conv(h, training)).permute(0, 2, 3, 1).reshape(h.shape[0], 180)
        logits = self.policy_fc(p)
        v = F.relu(self.value_conv(h, training)).permute(0, 2, 3, 1).reshape(h.shape[0], 90)
        v = torch.tanh(self.value_fc2(F.relu(self.value_fc1(v))))
        return logits, v

    # -- reference (TF) layout ----------------------------------------------------------------------
    def export_tf_layout(self):
        d = {}
        for i, cb in enumerate(self.convbns()):
            d["conv%d/kernel" % i] = cb.conv.weight.detach().float().permute(2, 3, 1, 0).contiguous().cpu().numpy().copy()  # HWIO
            d["conv%d/bias" % i] = cb.conv.bias.detach().float().cpu().numpy().copy()   # copies: never alias live parameters
            d["bn%d/moving_mean" % i] = cb.moving_mean.float().cpu().numpy().copy()
            d["bn%d/moving_variance" % i] = cb.moving_var.float().cpu().numpy().copy()
        for name, fc in (("policy_fc", self.policy_fc), ("value_fc1", self.value_fc1), ("value_fc2", self.value_fc2)):
            d[name + "/weights"] = fc.weight.detach().float().t().contiguous().cpu().numpy().copy()  # [in,out]
            d[name + "/biases"] = fc.bias.detach().float().cpu().numpy().copy()
        return d

    def load_tf_layout(self, d):
        with torch.no_grad():
            for i, cb in enumerate(self.convbns()):
                cb.conv.weight.copy_(torch.from_numpy(np.asarray(d["conv%d/kernel" % i])).permute(3, 2, 0, 1))
                cb.conv.bias.copy_(torch.from_numpy(np.asarray(d["conv%d/bias" % i])))
                cb.moving_mean.copy_(torch.from_numpy(np.asarray(d["bn%d/moving_mean" % i])))
                cb.moving_var.copy_(torch.from_numpy(np.asarray(d["bn%d/moving_variance" % i])))
            for name, fc in (("policy_fc", self.policy_fc), ("value_fc1", self.value_fc1), ("value_fc2", self.value_fc2)):
                fc.weight.copy_(torch.from_numpy(np.asarray(d[name + "/weights"])).t())
                fc.bias.copy_(torch.from_numpy(np.asarray(d[name + "/biases"])))


def tf_variable_names(res_block_nums):
    """Names of the reference graph's variables in a TF1 checkpoint -> keys of export_tf_layout()/load_tf_layout().

    The reference builds its graph without variable scopes (tf.name_scope does not rename variables), so TF1 numbers the
    layers in creation order (policy_value_network.py:45-74,151-162): `conv2d`, `conv2d_1`, ... — input conv, two per
    residual block, policy head conv, value head conv —, `BatchNorm`, `BatchNorm_1`, ... in the same order (no beta:
    center=False, no gamma: scale defaults to False), `fully_connected` (policy FC), `fully_connected_1`,
    `fully_connected_2` (value FCs).  Optimizer slots (`<var>/Momentum`) and `global_step` are not weights."""
    n_conv = 1 + 2 * res_block_nums + 2
    names = {}
    for i in range(n_conv):
        sfx = "" if i == 0 else "_%d" % i
        names["conv2d%s/kernel" % sfx] = "conv%d/kernel" % i
        names["conv2d%s/bias" % sfx] = "conv%d/bias" % i
        names["BatchNorm%s/moving_mean" % sfx] = "bn%d/moving_mean" % i
        names["BatchNorm%s/moving_variance" % sfx] = "bn%d/moving_variance" % i
    for j, ours in enumerate(("policy_fc", "value_fc1", "value_fc2")):
        sfx = "" if j == 0 else "_%d" % j
        names["fully_connected%s/weights" % sfx] = ours + "/weights"
        names["fully_connected%s/biases" % sfx] = ours + "/biases"
    return names


def from_tf_variables(variables, res_block_nums=None):
    """dict / NpzFile keyed by TF1 variable names (optionally with a ':0' suffix, Momentum slots and global_step
    present) -> (dict for load_tf_layout, res_block_nums, global_step or None).  The block count is inferred from the
    number of conv2d kernels when not given.  Shapes are checked against the reference graph."""
    v = {}
    for k in (variables.files if hasattr(variables, "files") else variables.keys()):
        name = k[:-2] if k.endswith(":0") else k
        v[name] = np.asarray(variables[k])
    n_conv = sum(1 for k in v if k.startswith("conv2d") and k.endswith("/kernel"))
    blocks = (n_conv - 3) // 2
    if n_conv < 3 or 1 + 2 * blocks + 2 != n_conv:
        raise ValueError("not a cchess-zero checkpoint: %d conv2d kernels" % n_conv)
    if res_block_nums is not None and res_block_nums != blocks:
        raise ValueError("checkpoint has %d residual blocks, expected %d" % (blocks, res_block_nums))
    out = {}
    for tf_name, ours in tf_variable_names(blocks).items():
        if tf_name not in v:
            raise KeyError("variable %s missing from the checkpoint" % tf_name)
        out[ours] = v[tf_name]
    want = {"conv0/kernel": (3, 3, 14, FILTERS), "conv%d/kernel" % (n_conv - 2): (1, 1, FILTERS, 2),
            "conv%d/kernel" % (n_conv - 1): (1, 1, FILTERS, 1), "policy_fc/weights": (180, PROB_SIZE),
            "value_fc1/weights": (90, 256), "value_fc2/weights": (256, 1)}
    for i in range(1, n_conv - 2):
        want["conv%d/kernel" % i] = (3, 3, FILTERS, FILTERS)
    for k, shp in want.items():
        if tuple(out[k].shape) != shp:
            raise ValueError("%s has shape %s, the reference graph has %s" % (k, tuple(out[k].shape), shp))
    gs = int(np.asarray(v["global_step"]).reshape(-1)[0]) if "global_step" in v else None
    return out, blocks, gs


def momentum_slots_from_tf_variables(variables, res_block_nums):
    """The MomentumOptimizer slot variables of a checkpoint (`<variable>/Momentum`, what tf.train.Saver persists next to the
    weights) -> dict keyed like export_tf_layout() (TF layout), or {} when the checkpoint holds none.  All or nothing."""
    v = {}
    for k in (variables.files if hasattr(variables, "files") else variables.keys()):
        name = k[:-2] if k.endswith(":0") else k
        v[name] = variables[k]
    out = {}
    for tf_name, ours in tf_variable_names(res_block_nums).items():
        if tf_name.split("/")[-1] in ("moving_mean", "moving_variance"):
            continue
        if tf_name + "/Momentum" not in v:
            return {}
        out[ours] = np.asarray(v[tf_name + "/Momentum"])
    return out


def to_tf_variables(module, global_step=None):
    """The inverse: a dict keyed by the reference graph's TF1 variable names (np.savez(**d) gives a file that a
    three-line TF script can assign back into the reference's graph)."""
    d = module.export_tf_layout()
    out = {tf_name: d[ours] for tf_name, ours in tf_variable_names(module.res_block_nums).items()}
    if global_step is not None:
        out["global_step"] = np.asarray(int(global_step), np.int64)
    return out


# (Round 5 had MX_DEPTH_LIMIT = 8 here: precision "strict" used k_trunk_mx_c128 up to 8 residual blocks and k_trunk_split_c128
# beyond, a depth constant with a 2x margin on one synthetic weight family.  Since round 6 "strict" MEASURES: the ladder starts at
# mx6 at any depth and PolicyValueNet.strict_check decides — TF-default weights at 19 blocks measure 1.5e-5 and stay on mx6, the
# peaked trained-like set of the tests measures 1.0e-3 there and falls over to the three-MFMA engine.)


def _e2m3_codes(x):
    """already-scaled values -> 6-bit E2M3 codes (sign, 2 exponent bits, 3 mantissa bits): round to nearest even, saturating
    at +-7.5 — what v_cvt_scalef32_2xpk16_fp6_f32 does to the activations (tools/experiments/mx_probe_check.py)"""
    a = x.abs().clamp(max=7.5)
    e = torch.floor(torch.log2(a.clamp(min=1.0)))                  # 0, 1, 2: the binade; below 1 the format is subnormal
    q = (torch.round(a / torch.exp2(e - 3)) * torch.exp2(e - 3)).clamp(max=7.5)   # torch.round: half to even
    e = torch.floor(torch.log2(q.clamp(min=1.0)))                  # rounding may have carried into the next binade
    normal = q >= 1.0
    mant = torch.where(normal, (q / torch.exp2(e) - 1.0) * 8.0, q * 8.0).round().to(torch.int64)
    ebits = torch.where(normal, e.to(torch.int64) + 1, torch.zeros_like(mant))
    return (torch.signbit(x).to(torch.int64) << 5) | (ebits << 3) | mant


def _pack_fp6(codes):
    """[..., 32] six-bit codes -> [..., 24] uint8, slot i at bits 6 i .. 6 i + 5 (four slots = one 24-bit word = three bytes)"""
    c = codes.to(torch.int64).reshape(codes.shape[:-1] + (8, 4))
    w = c[..., 0] | (c[..., 1] << 6) | (c[..., 2] << 12) | (c[..., 3] << 18)
    return torch.stack([w & 255, (w >> 8) & 255, (w >> 16) & 255], dim=-1).reshape(codes.shape[:-1] + (24,)).to(torch.uint8)


def mx_pack_layers(w):
    """Folded fp32 conv weights [L][128 co][128 ci][3][3] -> the 36 slabs per layer of k_trunk_mx_c128 (include/cchess_hip.h:
    cz_net_trunk_mx), uint8 [L][36][16384], all layers in one pass (a refresh() is ~60 torch ops whatever the depth).  Per
    32-input-channel quarter of a tap: fp16 w_hi in the strict engine's hi layout, then per (co, half h) the fp6 block of the 16
    channels c_j = 32 quarter + 8 (j / 4) + 4 h + j % 4: slot 2j = q6(2^11 w_lo), slot 2j+1 = q6(w_hi) under the block scale
    2^(exponent(amax) - 2); the E8M0 byte handed to the MFMA has the 2^-11 folded in."""
    dev, L = w.device, w.shape[0]
    t = w.permute(0, 3, 4, 2, 1).reshape(L, 9, 4, 4, 8, FILTERS).permute(0, 1, 2, 3, 5, 4).contiguous()    # [L][tap][quarter][ci8][co][ci%8]
    hi_bytes = t.to(torch.float16).contiguous().view(torch.uint8).reshape(L, 9, 4, 8192)
    wq = w.permute(0, 3, 4, 1, 2).reshape(L, 9, FILTERS, 4, 4, 2, 4)           # [L][tap][co][quarter][q][h][i], ci = 32 Q + 8 q + 4 h + i
    whi = wq.to(torch.float16).float()
    wlo = (wq - whi).to(torch.float16).float() * 2048.0
    whi = whi.permute(0, 1, 3, 5, 2, 4, 6).reshape(L, 9, 4, 2, FILTERS, 16)    # [L][tap][quarter][h][co][j = 4 q + i]
    wlo = wlo.permute(0, 1, 3, 5, 2, 4, 6).reshape(L, 9, 4, 2, FILTERS, 16)
    amax = torch.maximum(whi.abs().amax(-1), wlo.abs().amax(-1))
    _, e = torch.frexp(amax.clamp(min=1e-30))
    byte = (e + 124).clamp(min=12, max=254)                                    # biased exponent(amax) - 2
    sc = torch.exp2((byte - 127).float()).unsqueeze(-1)
    slots = torch.stack([_e2m3_codes(wlo / sc), _e2m3_codes(whi / sc)], dim=-1).reshape(L, 9, 4, 2, FILTERS, 32)
    blk = _pack_fp6(slots)                                                     # [L][9][4][2][128][24]
    xb = blk[..., :16].reshape(L, 9, 4, 4096)
    yb = blk[..., 16:].reshape(L, 9, 4, 2048)
    sdw = torch.zeros((L, 9, 4, 2, FILTERS, 4), dtype=torch.uint8, device=dev)
    sdw[..., 0] = (byte - 11).to(torch.uint8)
    pad = torch.zeros((L, 9, 4, 1024), dtype=torch.uint8, device=dev)
    return torch.cat([hi_bytes, xb, yb, sdw.reshape(L, 9, 4, 1024), pad], dim=-1).reshape(L, 36, 16384).contiguous()


def mx_pack_layer(w):
    """One layer [128 co][128 ci][3][3] -> uint8 [36][16384] (mx_pack_layers)."""
    return mx_pack_layers(w.unsqueeze(0))[0]


STRICT_CHECK_TOL = 5e-4        # precision "strict": the engine's MEASURED |dlogit| / |dvalue| against fp32 on the live weights must
STRICT_CHECK_POSITIONS = 64    # stay below this on this many distinct positions (half of north_star's 1e-3), else the next engine
_STRICT_LADDER = ("mx6", "fp16x2", "fp32")
_probe_planes = {}


def strict_probe_planes(ctx):
    """The positions precision "strict" measures itself on: STRICT_CHECK_POSITIONS distinct seeded random-playout positions
    (rules.random_positions: K1 / K2 on the device, plies 0 .. 80, both sides to move) as [n,9,10,14] float32 encoder planes on
    the context's device; made once per device."""
    key = ctx.device.index or 0
    if key not in _probe_planes:
        from .rules import Rules, random_positions
        r = Rules(ctx)
        boards, side, _ = random_positions(r, 4 * STRICT_CHECK_POSITIONS, seed=20260930, max_ply=80)
        rows = torch.unique(torch.cat([boards, side.unsqueeze(1)], dim=1), dim=0)      # sorted: deterministic
        rows = rows[torch.linspace(0, rows.shape[0] - 1, STRICT_CHECK_POSITIONS, device=rows.device).long()]
        assert torch.unique(rows, dim=0).shape[0] == STRICT_CHECK_POSITIONS
        _probe_planes[key] = r.encode_planes(rows[:, :90].contiguous(), rows[:, 90].contiguous()).float()
    return _probe_planes[key]


class PolicyValueNet:
    """Inference engine around PolicyValueModule: BN folded, tower in `dtype` (bf16/fp16/fp32, fp32
    accumulate on MFMA), heads in fp32.  forward_device() is the device-to-device path the search
    loop uses; forward() has the reference signature (policy_value_network.forward)."""

    def __init__(self, res_block_nums=7, device="cuda:0", dtype=torch.float16, seed=0, module=None, backend="auto", ctx=None,
                 split=False):
        """backend: "hip"       = first conv + residual tower + head convs in ONE fused MFMA launch (cz_net_trunk_bf16 /
                                  cz_net_trunk_f16; split=True: cz_net_trunk_split), FC heads in cz_fc_heads_f32; dtype bf16 or fp16,
                    "hip-layer" = one fused conv launch per layer, cz_conv3x3_c128_bf16 (bf16 only),
                    "torch"     = tower convs by torch/MIOpen (any dtype; the fp32 parity path),
                    "auto"      = hip for fp16 / bf16 on a GPU, else torch.
        dtype: fp16 is the default — the same MFMA rate as bf16 on gfx950 with 11 instead of 8 mantissa bits per stored
        activation: 7 blocks stay within north_star's 1e-3 of the fp32 graph on TF-default weights (|dlogit| 1.2e-4) and
        within 1.1e-3 of the largest logit on peaked, trained-like weights (bf16: 9e-3; tests/test_net.py).
        split: the STRICT engines (hip backend only).  True / "x3" = k_trunk_split_c128: every weight and every stored
        activation is carried as hi + lo, two values of `dtype` (22 significant bits for fp16, 16 for bf16), three MFMAs per
        product — north_star's 1e-3 against the fp32 graph also on peaked, trained-like weights and at 19 blocks (measured
        2e-5 / 1e-4), at a third of the 16-bit engine's rate.  "mx" = k_trunk_mx_c128 (fp16 only): the hi halves on fp16 MFMAs,
        both cross terms of the split on one block-scaled fp6 MFMA: 1.5 MFMA-equivalents per product, half the 16-bit engine's
        rate, 5e-4 / 2e-4 at 7 blocks (1.0e-3 / 1.1e-3 at 19).  "strict" = the ladder mx6 -> fp16x2 -> fp32, measured (strict_check)."""
        self.device = torch.device(device)
        self.dtype = dtype
        # split: False | True (= "x3": k_trunk_split_c128, three MFMAs per product) | "mx" (k_trunk_mx_c128: fp16 hi halves +
        # both cross terms on one block-scaled fp6 MFMA, 1.5 MFMA-equivalents per product, fp16 only) | "strict" (the cheapest
        # engine that MEASURES within STRICT_CHECK_TOL of fp32 on the live weights)
        # "strict" is a GUARANTEE, not a depth constant (round 6): the depth rule picks where to start, then every refresh() —
        # construction, restore(), each train_step — is followed (lazily, at the next evaluation) by strict_check(): the engine
        # is measured against the fp32 module on the live weights and the net falls over mx6 -> fp16x2 -> fp32 above
        # STRICT_CHECK_TOL.  strict_report holds the last measurement.
        self.strict_auto = split == "strict"
        self.strict_report = None
        self._check_pending = False
        self._fp32_fallback = False
        if split == "strict":
            split = "mx" if dtype == torch.float16 else True
        if split == "x3":
            split = True
        if split not in (False, True, "mx"):
            raise ValueError("split must be False, True / 'x3', 'mx' or 'strict'")
        self.mx = split == "mx"
        if self.mx and dtype != torch.float16:
            raise ValueError("the mx engine (k_trunk_mx_c128) computes with fp16 hi halves: dtype must be torch.float16")
        self.split = bool(split)
        self.module = (module or PolicyValueModule(res_block_nums, seed)).to(self.device)
        self.res_block_nums = self.module.res_block_nums
        if backend == "auto":
            backend = "hip" if (dtype in (torch.bfloat16, torch.float16) and self.device.type == "cuda") else "torch"
        if backend == "hip" and dtype not in (torch.bfloat16, torch.float16):
            raise ValueError("the fused hip backend computes in bf16 or fp16 (fp32 accumulate); use backend='torch' for %s" % dtype)
        if backend == "hip-layer" and dtype != torch.bfloat16:
            raise ValueError("the per-layer hip backend computes in bf16; use backend='hip' or 'torch' for %s" % dtype)
        if self.split and backend != "hip":
            raise ValueError("split=True (the strict engine) is a kernel of the fused hip backend")
        self.backend = backend
        self._ctx = ctx
        self._bufs = None
        self.conv_events = None  # set to a list to collect (start, end) HIP events around each conv launch
        self.fuse_policy_fc = True  # search loop: policy FC for the legal moves only, inside the expansion kernel
        self.refresh()

    @torch.no_grad()
    def refresh(self):
        """Re-fold BN and re-cast after a weight change.  Precision "strict": the engine ladder starts over and the new
        weights are measured at the next evaluation (strict_check)."""
        if self.strict_auto:
            self._strict_select(0 if self.dtype == torch.float16 else 1)
            self._check_pending = self.device.type == "cuda"
        self._pack()

    def _strict_select(self, rung):
        """engine of rung `rung` of the strict ladder: mx6 (k_trunk_mx_c128), fp16x2 (k_trunk_split_c128), fp32 (torch/MIOpen)"""
        self._rung = rung
        self.mx, self.split, self._fp32_fallback = rung == 0, rung in (0, 1), rung == 2
        self.backend = "torch" if rung == 2 else "hip"

    @property
    def engine_name(self):
        """Which arithmetic evaluates the tower right now: mx6 | fp16x2 | bf16x2 | fp16 | bf16 | fp32."""
        if self._fp32_fallback or self.dtype == torch.float32:
            return "fp32"
        half = "fp16" if self.dtype == torch.float16 else "bf16"
        return "mx6" if self.mx else (half + "x2" if self.split else half)

    @torch.no_grad()
    def strict_check(self, planes_nhwc=None, tol=None):
        """Precision "strict" measured on the LIVE weights (policy_value_network.py:202-214 is an fp32 sess.run; north_star
        allows 1e-3): the selected engine and the fp32 torch module evaluate the same >= 64 distinct positions; above `tol`
        (default STRICT_CHECK_TOL = 5e-4: a factor of two inside the contract, the probe is 64 positions and not every
        position) the net falls over to the next engine of the ladder mx6 -> fp16x2 -> fp32 and measures again.
        -> strict_report = {"engine", "dlogit", "dvalue", "max_abs_logit", "tol", "positions", "fell_over_from": [...]}."""
        import warnings
        tol = STRICT_CHECK_TOL if tol is None else float(tol)
        self._check_pending = False
        x = strict_probe_planes(self._hip_ctx()) if planes_nhwc is None else torch.as_tensor(np.asarray(planes_nhwc.cpu() if torch.is_tensor(planes_nhwc) else planes_nhwc, dtype=np.float32)).to(self.device)
        lr, vr = self.module.float()(x.permute(0, 3, 1, 2).contiguous())
        lr, vr = lr.double(), vr.double().reshape(-1)
        failed = []
        while True:
            l, v = self.forward_device(x)
            dl = float((l.double() - lr).abs().max())
            dv = float((v.double().reshape(-1) - vr).abs().max())
            ok = dl <= tol and dv <= tol      # NaN compares false: a non-finite engine fails
            rep = {"engine": self.engine_name, "dlogit": dl, "dvalue": dv, "max_abs_logit": float(lr.abs().max()), "tol": tol,
                   "positions": int(x.shape[0]), "fell_over_from": list(failed)}
            if ok or not self.strict_auto or self._rung >= 2:
                break
            failed.append(dict(engine=self.engine_name, dlogit=dl, dvalue=dv))
            warnings.warn("precision 'strict': %s measures |dlogit| %.3g / |dvalue| %.3g against fp32 on these weights (> %.1g, |logit| <= %.3g): "
                          "falling over to %s" % (self.engine_name, dl, dv, tol, rep["max_abs_logit"], _STRICT_LADDER[self._rung + 1]))
            self._strict_select(self._rung + 1)
            self._pack()
        self.strict_report = rep
        return rep

    def _ensure_checked(self):
        if self._check_pending:
            self.strict_check()

    @torch.no_grad()
    def _pack(self):
        """the operand images of the selected engine"""
        m, dt = self.module, (torch.float32 if self._fp32_fallback else self.dtype)
        cl = torch.channels_last

        def conv_pack(cb):
            w, b = cb.folded()
            return w.to(dt).contiguous(memory_format=cl), b.to(dt)
        self.w_in = conv_pack(m.conv_in)
        self.w_blocks = [(conv_pack(a), conv_pack(b)) for a, b in m.blocks]
        if self.backend.startswith("hip"):
            hdt = torch.float16 if dt == torch.float16 else torch.bfloat16

            def hip_pack(cb):
                w, b = cb.folded()  # [O,I,3,3] fp32 with the BN scale folded in
                # -> [tap = dy*3+dx][ci/8][co][ci%8] bf16: the LDS image of the B operand, slab by slab
                wp = w.permute(2, 3, 1, 0).reshape(9, 16, 8, FILTERS).permute(0, 1, 3, 2).contiguous().to(hdt)
                return wp, b.float().contiguous()
            self.hip_blocks = [(hip_pack(a), hip_pack(b)) for a, b in m.blocks]
            # first layer 14 -> 128: input channels padded to 16, [tap][ci/8][co][ci%8] bf16
            w0, b0 = m.conv_in.folded()
            w0p = torch.zeros((3, 3, 16, FILTERS), dtype=torch.float32, device=w0.device)
            w0p[:, :, :14, :] = w0.permute(2, 3, 1, 0)
            self.hip_w0 = w0p.reshape(9, 2, 8, FILTERS).permute(0, 1, 3, 2).contiguous().to(hdt)
            self.hip_b0 = b0.float().contiguous()
            layers = [x for blk in self.hip_blocks for x in blk]
            self.hip_tower_w = torch.stack([w for w, _ in layers]).contiguous() if layers else torch.zeros((0,), dtype=hdt, device=self.device)
            self.hip_tower_b = torch.stack([b for _, b in layers]).contiguous() if layers else torch.zeros((0,), dtype=torch.float32, device=self.device)
            if self.mx:
                sl = [cb.folded()[0] for blk in m.blocks for cb in blk]
                self.hip_mx_w = mx_pack_layers(torch.stack(sl)) if sl else torch.zeros((0,), dtype=torch.uint8, device=self.device)
            if self.split:
                # strict engine: w = hi + lo, two values of hdt.  Tower layer: [tap][32-channel quarter of the tap = one 16 KB
                # slab][hi, lo][ci/8 within the quarter][co][ci%8]; first layer: [tap][hi, lo][ci/8][co][ci%8]
                def halves(w):
                    hi = w.to(hdt)
                    return hi, (w - hi.float()).to(hdt)

                def split_pack(cb):
                    w, _ = cb.folded()
                    t = w.permute(2, 3, 1, 0).reshape(9, 4, 4, 8, FILTERS).permute(0, 1, 2, 4, 3)   # [tap][quarter][ci8][co][ci%8]
                    hi, lo = halves(t)
                    return torch.stack([hi, lo], dim=2).contiguous()                                 # [9][4][2][4][128][8]
                sl = [] if self.mx else [split_pack(cb) for blk in m.blocks for cb in blk]
                self.hip_split_w = torch.stack(sl).contiguous() if sl else torch.zeros((0,), dtype=hdt, device=self.device)
                hi0, lo0 = halves(w0p.reshape(9, 2, 8, FILTERS).permute(0, 1, 3, 2))
                self.hip_split_w0 = torch.stack([hi0, lo0], dim=1).contiguous()                      # [9][2][2][128][8]
        # heads: 1x1 convs as fp32 matmuls over [B*90,128]
        wp, bp = m.policy_conv.folded()
        wv, bv = m.value_conv.folded()
        self.head_w_rows = torch.cat([wp.view(2, FILTERS), wv.view(1, FILTERS)], 0).contiguous()  # [3,128]
        self.head_w = self.head_w_rows.t().contiguous()  # [128,3]
        self.head_b = torch.cat([bp, bv], 0).contiguous()
        self.pfc_w = m.policy_fc.weight.float().t().contiguous()
        self.pfc_b = m.policy_fc.bias.float()
        self.v1_w = m.value_fc1.weight.float().t().contiguous()
        self.v1_b = m.value_fc1.bias.float()
        self.v2_w = m.value_fc2.weight.float().t().contiguous()
        self.v2_b = m.value_fc2.bias.float()
        if self.backend.startswith("hip"):
            # policy FC weight [2086,180] split into bf16 hi + lo and packed in MFMA B-fragment order
            # [label/32][k/16][lane = (k%16)/8*32 + label%32][k%8] (cz_fc_heads_f32)
            wpad = torch.zeros((66 * 32, 192), dtype=torch.float32, device=self.device)
            wpad[:PROB_SIZE, :180] = m.policy_fc.weight.float()
            hi = wpad.to(torch.bfloat16)
            lo = (wpad - hi.float()).to(torch.bfloat16)

            def frag(w):
                return w.view(66, 32, 12, 2, 8).permute(0, 2, 3, 1, 4).contiguous()
            self.hip_pfc_hi, self.hip_pfc_lo = frag(hi), frag(lo)
            self.hip_pfc_b = m.policy_fc.bias.detach().float().contiguous()
            self.pfc_w_rows = m.policy_fc.weight.detach().float().contiguous()   # [2086,180]: rows the expansion kernel gathers
            self.pfc_b_f32 = self.hip_pfc_b
            self.hip_v1_wt = m.value_fc1.weight.float().t().contiguous()      # [90,256]
            self.hip_v1_b = m.value_fc1.bias.float().contiguous()
            self.hip_v2_w = m.value_fc2.weight.float().reshape(-1).contiguous()  # [256]
            self.hip_v2_b = m.value_fc2.bias.float().contiguous()

    def _hip_ctx(self):
        if self._ctx is None:
            from .engine import Context
            self._ctx = Context(1, 2, self.device.index or 0)
        return self._ctx

    def _hip_conv(self, x, wb, res, out, relu=True):
        """x, res, out: [B,90,128] bf16 contiguous device tensors; one launch = one fused layer."""
        import ctypes as C
        from ._lib import check, lib
        wp, b = wb
        ev = None
        if self.conv_events is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        check(lib().cz_conv3x3_c128_bf16(self._hip_ctx().h, C.c_void_p(x.data_ptr()), C.c_void_p(wp.data_ptr()),
                                         C.c_void_p(b.data_ptr()), C.c_void_p(res.data_ptr()) if res is not None else None,
                                         C.c_void_p(out.data_ptr()), x.shape[0], 1 if relu else 0), "cz_conv3x3_c128_bf16")
        if ev is not None:
            ev[1].record()
            self.conv_events.append(ev)
        return out

    def _hip_net_forward(self, planes, trunk=False):
        """planes [B,9,10,C] (C = 16 in the net's 16-bit dtype: zero-copy; anything else is repacked) -> z [B,90,3] f32
        (post-ReLU head conv outputs), or with trunk=True the trunk activations [B,90,128] in the net's dtype.
        First conv + residual tower + head 1x1 convs in ONE launch (cz_net_trunk_bf16 / cz_net_trunk_f16)."""
        import ctypes as C
        from ._lib import check, lib
        B = planes.shape[0]
        hdt = torch.float16 if self.dtype == torch.float16 else torch.bfloat16
        fn = lib().cz_net_trunk_f16 if hdt == torch.float16 else lib().cz_net_trunk_bf16
        if planes.dtype == hdt and planes.shape[-1] == 16 and planes.is_contiguous():
            p16 = planes
        else:
            p16 = torch.zeros((B, 9, 10, 16), dtype=hdt, device=self.device)
            p16[..., :14] = planes[..., :14].to(hdt)
        tdt = torch.float32 if self.split else hdt     # the strict engine hands the trunk out as fp32 = hi + lo
        z = torch.empty((B, 90, FILTERS), dtype=tdt, device=self.device) if trunk else torch.empty((B, 90, 3), dtype=torch.float32, device=self.device)
        self._hip_ctx().bind_stream()
        ev = None
        if self.conv_events is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        if self.mx:
            check(lib().cz_net_trunk_mx(self._hip_ctx().h, C.c_void_p(p16.data_ptr()), C.c_void_p(self.hip_split_w0.data_ptr()),
                                        C.c_void_p(self.hip_b0.data_ptr()), C.c_void_p(self.hip_mx_w.data_ptr()),
                                        C.c_void_p(self.hip_tower_b.data_ptr()), C.c_void_p(z.data_ptr()) if trunk else None,
                                        C.c_void_p(self.head_w_rows.data_ptr()), C.c_void_p(self.head_b.data_ptr()),
                                        None if trunk else C.c_void_p(z.data_ptr()), B, self.res_block_nums), "cz_net_trunk_mx")
        elif self.split:
            check(lib().cz_net_trunk_split(self._hip_ctx().h, C.c_void_p(p16.data_ptr()), C.c_void_p(self.hip_split_w0.data_ptr()),
                                           C.c_void_p(self.hip_b0.data_ptr()), C.c_void_p(self.hip_split_w.data_ptr()),
                                           C.c_void_p(self.hip_tower_b.data_ptr()), C.c_void_p(z.data_ptr()) if trunk else None,
                                           C.c_void_p(self.head_w_rows.data_ptr()), C.c_void_p(self.head_b.data_ptr()),
                                           None if trunk else C.c_void_p(z.data_ptr()), B, self.res_block_nums,
                                           2 if hdt == torch.float16 else 1), "cz_net_trunk_split")
        else:
            check(fn(self._hip_ctx().h, C.c_void_p(p16.data_ptr()), C.c_void_p(self.hip_w0.data_ptr()),
                     C.c_void_p(self.hip_b0.data_ptr()), C.c_void_p(self.hip_tower_w.data_ptr()),
                     C.c_void_p(self.hip_tower_b.data_ptr()), C.c_void_p(z.data_ptr()) if trunk else None,
                     C.c_void_p(self.head_w_rows.data_ptr()), C.c_void_p(self.head_b.data_ptr()),
                     None if trunk else C.c_void_p(z.data_ptr()), B, self.res_block_nums), "cz_net_trunk")
        if ev is not None:
            ev[1].record()
            self.conv_events.append(ev)
        return z

    def _hip_tower_heads_forward(self, h):
        """h: [B,128,9,10] channels_last bf16 -> z [B,90,3] f32: residual tower + both head 1x1 convs (+BN+ReLU) in
        one launch; the trunk never leaves the CU."""
        import ctypes as C
        from ._lib import check, lib
        B = h.shape[0]
        if not h.is_contiguous(memory_format=torch.channels_last):
            h = h.contiguous(memory_format=torch.channels_last)
        x = h.permute(0, 2, 3, 1).reshape(B, 90, FILTERS)
        z = torch.empty((B, 90, 3), dtype=torch.float32, device=self.device)
        self._hip_ctx().bind_stream()
        ev = None
        if self.conv_events is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        check(lib().cz_tower_heads_c128_bf16(self._hip_ctx().h, C.c_void_p(x.data_ptr()), C.c_void_p(self.hip_tower_w.data_ptr()),
                                             C.c_void_p(self.hip_tower_b.data_ptr()), None, C.c_void_p(self.head_w_rows.data_ptr()),
                                             C.c_void_p(self.head_b.data_ptr()), C.c_void_p(z.data_ptr()), B, self.res_block_nums),
              "cz_tower_heads_c128_bf16")
        if ev is not None:
            ev[1].record()
            self.conv_events.append(ev)
        return z

    def _hip_tower_forward(self, h):
        """h: [B,128,9,10] channels_last bf16 -> all residual blocks in one launch (activations stay in LDS)."""
        import ctypes as C
        from ._lib import check, lib
        B = h.shape[0]
        if not h.is_contiguous(memory_format=torch.channels_last):
            h = h.contiguous(memory_format=torch.channels_last)
        x = h.permute(0, 2, 3, 1).reshape(B, 90, FILTERS)
        if self.res_block_nums == 0:
            return h
        self._hip_ctx().bind_stream()
        ev = None
        if self.conv_events is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        check(lib().cz_tower_c128_bf16(self._hip_ctx().h, C.c_void_p(x.data_ptr()), C.c_void_p(self.hip_tower_w.data_ptr()),
                                       C.c_void_p(self.hip_tower_b.data_ptr()), C.c_void_p(x.data_ptr()), B, self.res_block_nums),
              "cz_tower_c128_bf16")
        if ev is not None:
            ev[1].record()
            self.conv_events.append(ev)
        return x.reshape(B, 9, 10, FILTERS).permute(0, 3, 1, 2)

    def _hip_blocks_forward(self, h):
        """h: [B,128,9,10] channels_last bf16 (first conv output) -> same shape after all residual blocks."""
        B = h.shape[0]
        if not h.is_contiguous(memory_format=torch.channels_last):
            h = h.contiguous(memory_format=torch.channels_last)
        cur = h.permute(0, 2, 3, 1).reshape(B, 90, FILTERS)  # same memory viewed as NHWC rows
        if self._bufs is None or self._bufs[0].shape[0] != B:
            self._bufs = [torch.empty((B, 90, FILTERS), dtype=torch.bfloat16, device=self.device) for _ in range(3)]
        self._hip_ctx().bind_stream()
        t = self._bufs[0]
        for i, (w1, w2) in enumerate(self.hip_blocks):
            y = self._bufs[1 + (i & 1)]           # never the buffer `cur` lives in
            self._hip_conv(cur, w1, None, t)      # conv + BN + ReLU
            self._hip_conv(t, w2, cur, y)         # conv + BN + residual + ReLU
            cur = y
        return cur.reshape(B, 9, 10, FILTERS).permute(0, 3, 1, 2)

    @torch.no_grad()
    def first_conv(self, planes):
        """planes [B,9,10,C>=14] (NHWC, any float dtype) -> conv3x3(14->128)+BN+ReLU, [B,128,9,10] channels_last."""
        x = planes[..., :14] if planes.shape[-1] != 14 else planes
        x = x.to(self.w_in[0].dtype).permute(0, 3, 1, 2)  # NHWC memory == NCHW channels_last view (fp32 on the strict ladder's last rung)
        if not x.is_contiguous(memory_format=torch.channels_last):
            x = x.contiguous(memory_format=torch.channels_last)
        return F.relu_(F.conv2d(x, self.w_in[0], self.w_in[1], padding=1))

    @torch.no_grad()
    def tower(self, planes):
        """planes [B,9,10,C>=14] (NHWC, any float dtype) -> trunk activations [B,128,9,10] channels_last."""
        self._ensure_checked()
        if self.backend == "hip" and (self.dtype == torch.float16 or self.split) and self.res_block_nums >= 1:
            t = self._hip_net_forward(planes, trunk=True)   # fp16 / strict: the fused kernel is the only hip route
            return t.reshape(t.shape[0], 9, 10, FILTERS).permute(0, 3, 1, 2)
        h = self.first_conv(planes)
        if self.backend == "hip":
            return self._hip_tower_forward(h)
        if self.backend == "hip-layer":
            return self._hip_blocks_forward(h)
        for (w1, b1), (w2, b2) in self.w_blocks:
            t = F.relu_(F.conv2d(h, w1, b1, padding=1))
            h = F.relu_(F.conv2d(t, w2, b2, padding=1).add_(h))
        return h

    @torch.no_grad()
    def heads(self, h):
        B = h.shape[0]
        hw = h.permute(0, 2, 3, 1).reshape(B * 90, FILTERS).float()   # (b,h,w) rows, NHWC order
        z = torch.relu_(torch.addmm(self.head_b, hw, self.head_w))    # [B*90,3]: 2 policy + 1 value channel
        p = z[:, :2].reshape(B, 180)                                   # flatten in (h,w,c) order
        logits = torch.addmm(self.pfc_b, p, self.pfc_w)
        v = z[:, 2].reshape(B, 90)
        v = torch.relu_(torch.addmm(self.v1_b, v, self.v1_w))
        v = torch.tanh(torch.addmm(self.v2_b, v, self.v2_w))
        return logits, v

    @torch.no_grad()
    def fc_heads(self, z):
        """z [B,90,3] f32 (post-ReLU head conv outputs) -> (logits, value): the three FC layers."""
        B = z.shape[0]
        p = z[:, :, :2].reshape(B, 180)                                # (h,w,c) flatten, policy_value_network.py:62
        logits = torch.addmm(self.pfc_b, p, self.pfc_w)
        v = z[:, :, 2]
        v = torch.relu_(torch.addmm(self.v1_b, v, self.v1_w))
        v = torch.tanh(torch.addmm(self.v2_b, v, self.v2_w))
        return logits, v

    def _hip_fc_heads(self, z):
        """z [B,90,3] f32 -> (logits [B,2086] f32, value [B,1] f32): policy FC on MFMA (split-bf16, fp32
        accumulate) + value FCs on the fp32 VALU, cz_fc_heads_f32."""
        import ctypes as C
        from ._lib import check, lib
        B = z.shape[0]
        logits = torch.empty((B, PROB_SIZE), dtype=torch.float32, device=self.device)
        value = torch.empty((B, 1), dtype=torch.float32, device=self.device)
        self._hip_ctx().bind_stream()
        vp = lambda t: C.c_void_p(t.data_ptr())
        check(lib().cz_fc_heads_f32(self._hip_ctx().h, vp(z), vp(self.hip_pfc_hi), vp(self.hip_pfc_lo), vp(self.hip_pfc_b),
                                    vp(self.hip_v1_wt), vp(self.hip_v1_b), vp(self.hip_v2_w), vp(self.hip_v2_b),
                                    vp(logits), vp(value), B), "cz_fc_heads_f32")
        return logits, value

    @torch.no_grad()
    def range_check(self, planes_nhwc=None, raise_on_nonfinite=True):
        """After new weights (restore()): evaluate the start position (or the given planes) and look at what the engine makes
        of them.  The fp16 kernels saturate at 65504 instead of producing inf (k_tower8_c128: packed min on the store;
        k_trunk_split_c128: clamp before the split), so a checkpoint with huge activations cannot poison the priors with
        NaN — but a saturated tower is no longer the reference's function: warn when the trunk comes within a factor of two
        of the half range (use precision "bf16x2" / "fp32" for such weights), raise when an output is not finite.
        -> {"max_activation", "finite"}."""
        import warnings
        if planes_nhwc is None:
            x = torch.zeros((1, 9, 10, 14), dtype=torch.float32, device=self.device)
            x[0, 0, 4, 0] = 1.0   # any plane pattern will do; the check is about the weights' scale
            x[0, 8, 4, 7] = 1.0
        else:
            x = torch.as_tensor(np.asarray(planes_nhwc, dtype=np.float32)).to(self.device)
        logits, v = self.forward_device(x)
        finite = bool(torch.isfinite(logits).all()) and bool(torch.isfinite(v).all())
        peak = float(self.tower(x).float().abs().max())
        if self.dtype == torch.float16 and peak > 32752.0:
            warnings.warn("trunk activations reach %.3g: the fp16 engine saturates at 65504; use precision 'bf16x2' or 'fp32' for these weights" % peak)
        if not finite and raise_on_nonfinite:
            raise FloatingPointError("the net's outputs are not finite for these weights (peak trunk activation %.3g)" % peak)
        return {"max_activation": peak, "finite": finite}

    @property
    def fused_search(self):
        """True when the search loop may skip the full policy FC (SearchEngine.step -> expand_backup_fc)."""
        self._ensure_checked()   # precision "strict": new weights are measured before the loop decides how to call the net
        return self.backend == "hip" and self.res_block_nums >= 1 and self.fuse_policy_fc

    @torch.no_grad()
    def search_eval(self, planes, n_rows=None):
        """Device planes -> (z [B,90,3] f32 head-conv outputs, value [B,1] f32): what the search needs when the policy
        FC is evaluated inside the expansion kernel (cz_search_expand_backup_fc).  n_rows: device address of an int
        (SearchEngine.select_compact) — only the first *n_rows rows are computed, the rest of z / value is undefined."""
        import ctypes as C
        from ._lib import check, lib
        self._ensure_checked()
        h = self._hip_ctx().h
        if n_rows is not None:
            check(lib().cz_set_batch_count(h, n_rows), "cz_set_batch_count")
        try:
            z = self._hip_net_forward(planes)
            B = z.shape[0]
            value = torch.empty((B, 1), dtype=torch.float32, device=self.device)
            vp = lambda t: C.c_void_p(t.data_ptr())
            check(lib().cz_fc_heads_f32(h, vp(z), None, None, None, vp(self.hip_v1_wt), vp(self.hip_v1_b),
                                        vp(self.hip_v2_w), vp(self.hip_v2_b), None, vp(value), B), "cz_fc_heads_f32")
        finally:
            if n_rows is not None:
                check(lib().cz_set_batch_count(h, None), "cz_set_batch_count")
        return z, value

    @torch.no_grad()
    def forward_device(self, planes):
        """Device planes [B,9,10,C] -> (logits [B,2086] f32, value [B,1] f32), all on the device."""
        self._ensure_checked()
        if self.backend == "hip" and self.res_block_nums >= 1:
            return self._hip_fc_heads(self._hip_net_forward(planes))
        return self.heads(self.tower(planes))

    @torch.no_grad()
    def forward(self, positions):
        """Reference signature (policy_value_network.py:202-214): ndarray or list of [9,10,14] ->
        (logits [B,2086] float32 ndarray, value [B,1] float32 ndarray)."""
        x = torch.as_tensor(np.asarray(positions, dtype=np.float32)).to(self.device)
        if x.dim() == 3:
            x = x.unsqueeze(0)
        logits, v = self.forward_device(x)
        return logits.cpu().numpy(), v.cpu().numpy()


def trained_like_(net, planes_nhwc, seed=5):
    """A weight set with the statistics of a trained net rather than of Glorot noise (used by the net-error block of
    bench.py and by tests/nethelpers.py): non-negative policy FC weights with a few strong feature -> move links (raw
    logits are the priors of this engine — quirk Q3 — and a trained net's are positive and peaked on a few moves:
    |logit| ~ 10, softmax far from uniform), a value head with spread, non-trivial BN statistics and biases.
    planes_nhwc: [n,9,10,14] float32 encoder outputs (ndarray or tensor) the two heads are calibrated on."""
    gen = torch.Generator().manual_seed(seed)
    m = net.module
    with torch.no_grad():
        for cb in m.convbns():
            dev = cb.conv.bias.device
            cb.conv.bias.copy_((torch.randn(cb.conv.bias.shape, generator=gen) * 0.05).to(dev))
            cb.moving_var.copy_((torch.rand(cb.moving_var.shape, generator=gen) * 0.5 + 0.75).to(dev))
        w = m.policy_fc.weight
        peaked = (torch.rand(w.shape, generator=gen) < 0.06).to(w.device)
        w.copy_(w.abs() + peaked * w.abs() * 15.0)
        m.policy_fc.bias.copy_((torch.rand(2086, generator=gen) * 0.02).to(w.device))
        # both heads are calibrated on the given positions, whatever the depth of the tower: the policy FC is scaled so
        # that the largest logit of a position averages 10; the last value layer so that tanh's argument has mean 0 and
        # std 0.6 (a Glorot-initialised head answers ~-0.65 +- 0.03 for every position: no value signal for a search)
        x = torch.as_tensor(np.asarray(planes_nhwc.cpu() if torch.is_tensor(planes_nhwc) else planes_nhwc, dtype=np.float32)).to(w.device).permute(0, 3, 1, 2)
        feats, vconv = [], []
        hook = m.value_fc2.register_forward_hook(lambda mod, inp, out: feats.append(inp[0].detach()))
        hook2 = m.value_conv.register_forward_hook(lambda mod, inp, out: vconv.append(out.detach()))
        logits, _ = m(x)
        w.mul_(10.0 / float(logits.max(dim=1).values.mean()))
        # a value head whose 1x1 conv is negative on (nearly) every cell of every position is DEAD behind its ReLU — it happens
        # at 19 blocks, where the trunk's activations have drifted — and would answer one constant: dvalue = 0 says nothing
        # about an engine.  Its bias is shifted to the median of the pre-activations (half of the cells fire), then the two
        # FC layers see a signal to calibrate on.
        if float((vconv[0] > 0).float().mean()) < 0.1:
            m.value_conv.conv.bias.sub_(vconv[0].median() * torch.sqrt(m.value_conv.moving_var + BN_EPS))
            del feats[:], vconv[:]
            m(x)
        hook.remove()
        hook2.remove()
        pre = feats[0] @ m.value_fc2.weight.t()
        sd = float(pre.std()) if pre.shape[0] > 1 else 0.0
        if not sd > 1e-5 * max(1.0, float(pre.abs().mean())):     # identical rows differ by rounding noise at most
            raise ValueError("trained_like_: the value head answers one constant on the %d calibration positions (identical positions?): "
                             "calibrate on distinct positions" % pre.shape[0])
        k = 0.6 / sd
        m.value_fc2.weight.mul_(k)
        m.value_fc2.bias.fill_(-float(pre.mean()) * k)
    net.refresh()
    return net


def net_error(net, planes_nhwc):
    """The 16-bit engine's deviation from fp32 on the SAME weights and inputs: (max |dlogit|, that relative to the largest
    |logit|, max |dsoftmax|, max |dvalue|, argmax agreement) of `net` (any engine) against the fp32 torch module evaluated
    on the device (the fp32 engine is itself held to <= 1e-3 — measured 1e-7 .. 3e-5 — of the NumPy restatement of the
    reference graph by tests/test_net.py).  planes_nhwc: [n,9,10,14] float32 device tensor."""
    with torch.no_grad():
        x = planes_nhwc.to(net.device).float()
        l16, v16 = net.forward_device(x)
        m32 = net.module.float()
        lr, vr = m32(x.permute(0, 3, 1, 2).contiguous())
        l16, v16, lr, vr = l16.double(), v16.double().reshape(-1), lr.double(), vr.double().reshape(-1)
        ml = float(lr.abs().max())
        dl = float((l16 - lr).abs().max())
        return {"positions": int(x.shape[0]), "max_abs_logit": ml, "dlogit": dl, "dlogit_rel": dl / ml,
                "dsoftmax": float((torch.softmax(l16, 1) - torch.softmax(lr, 1)).abs().max()),
                "max_softmax": float(torch.softmax(lr, 1).max()),
                "dvalue": float((v16 - vr).abs().max()),
                "argmax_agree": float((l16.argmax(1) == lr.argmax(1)).double().mean())}


def flops_per_position(res_block_nums):
    """2 x MACs with SAME-padding taps counted densely."""
    conv_in = 90 * 9 * 14 * 128
    res = 90 * 9 * 128 * 128
    heads = 90 * 128 * 2 + 180 * 2086 + 90 * 128 + 90 * 256 + 256
    return 2 * (conv_in + 2 * res_block_nums * res + heads)
