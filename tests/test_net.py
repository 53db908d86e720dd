"""Net parity.  CPU: the torch module graph == the NumPy restatement of the TF graph (fp32).
GPU: the inference engine (BN folded, channels_last, MFMA convs) within 1e-3 of the fp32
restatement in fp32 mode; bf16 mode checked on softmax probabilities and value."""
import os

import numpy as np
import pytest
import torch

import nethelpers as H
from oracle import net_numpy

_positions = H.positions


@pytest.mark.parametrize("blocks", [1, 2])
def test_module_matches_numpy_restatement_cpu(blocks):
    from cchess_zero_amd.net import PolicyValueModule, flops_per_position
    assert flops_per_position(7) == 375358832 and flops_per_position(19) == 2 * 506184376
    m = PolicyValueModule(blocks, seed=3)
    # non-trivial BN statistics and biases so that every term of the graph is exercised
    gen = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for cb in m.convbns():
            cb.moving_mean.copy_(torch.randn(cb.moving_mean.shape, generator=gen) * 0.1)
            cb.moving_var.copy_(torch.rand(cb.moving_var.shape, generator=gen) + 0.5)
            cb.conv.bias.copy_(torch.randn(cb.conv.bias.shape, generator=gen) * 0.1)
    x = _positions(6)
    with torch.no_grad():
        lt, vt = m(torch.from_numpy(x).permute(0, 3, 1, 2))
    ln, vn = net_numpy.forward(m.export_tf_layout(), x, blocks)
    assert np.abs(lt.numpy() - ln).max() < 1e-4
    assert np.abs(vt.numpy() - vn).max() < 1e-5
    # TF-layout round trip
    m2 = PolicyValueModule(blocks, seed=9)
    m2.load_tf_layout(m.export_tf_layout())
    with torch.no_grad():
        l2, v2 = m2(torch.from_numpy(x).permute(0, 3, 1, 2))
    assert torch.equal(l2, lt) and torch.equal(v2, vt)


def test_tf1_variable_names_roundtrip_cpu():
    """a cchess-zero TF1 checkpoint's variables by name (`conv2d_3/kernel`, `BatchNorm_3/moving_variance`,
    `fully_connected_1/weights` ... as tf.train.load_checkpoint lists them, with Momentum slots and global_step mixed in,
    ':0' suffixes allowed) load into the module; HWIO / [in,out] layouts are converted; shapes are checked."""
    from cchess_zero_amd.net import PolicyValueModule, from_tf_variables, tf_variable_names, to_tf_variables
    names = tf_variable_names(7)
    assert len(names) == 4 * 17 + 6 and names["conv2d/kernel"] == "conv0/kernel" and names["conv2d_16/bias"] == "conv16/bias"
    assert names["BatchNorm_15/moving_mean"] == "bn15/moving_mean" and names["fully_connected_2/weights"] == "value_fc2/weights"
    m = PolicyValueModule(2, seed=3)
    gen = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for cb in m.convbns():
            cb.moving_mean.copy_(torch.randn(cb.moving_mean.shape, generator=gen) * 0.1)
            cb.moving_var.copy_(torch.rand(cb.moving_var.shape, generator=gen) + 0.5)
    ck = {k + ":0": v for k, v in to_tf_variables(m, global_step=1234).items()}
    assert ck["conv2d/kernel:0"].shape == (3, 3, 14, 128) and ck["conv2d_5/kernel:0"].shape == (1, 1, 128, 2)
    assert ck["fully_connected/weights:0"].shape == (180, 2086) and ck["fully_connected_2/weights:0"].shape == (256, 1)
    for k in list(ck):                                  # what a real checkpoint also holds
        if k.endswith("kernel:0") or k.endswith("weights:0"):
            ck[k[:-2] + "/Momentum:0"] = np.zeros_like(ck[k])
    d, blocks, gs = from_tf_variables(ck)
    assert blocks == 2 and gs == 1234
    m2 = PolicyValueModule(2, seed=8)
    m2.load_tf_layout(d)
    x = torch.from_numpy(_positions(5, 1)).permute(0, 3, 1, 2)
    with torch.no_grad():
        a, b = m(x), m2(x)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    # and the NumPy restatement of the TF graph consumes the very same arrays
    ln, vn = net_numpy.forward(d, _positions(5, 1), 2)
    assert np.abs(a[0].numpy() - ln).max() < 1e-4 and np.abs(a[1].numpy() - vn).max() < 1e-5
    with pytest.raises(ValueError):
        from_tf_variables(ck, res_block_nums=7)
    bad = dict(ck)
    bad["conv2d_1/kernel:0"] = np.zeros((3, 3, 64, 128), np.float32)
    with pytest.raises(ValueError):
        from_tf_variables(bad)
    del bad["BatchNorm/moving_mean:0"]
    with pytest.raises((KeyError, ValueError)):
        from_tf_variables(bad)


@pytest.mark.gpu
@pytest.mark.parametrize("blocks", [2, 7])
def test_inference_engine_fp32_within_1e3(blocks):
    from cchess_zero_amd.net import PolicyValueNet
    net = PolicyValueNet(blocks, "cuda:0", torch.float32, seed=1)
    x = _positions(32, 1)
    logits, v = net.forward(x)
    ln, vn = net_numpy.forward(net.module.export_tf_layout(), x, blocks)
    assert logits.shape == (32, 2086) and v.shape == (32, 1) and logits.dtype == np.float32
    assert np.abs(logits - ln).max() < 1e-3   # north_star tolerance: 1e-3 fp32
    assert np.abs(v - vn).max() < 1e-3
    # list-of-arrays input, as policy_update passes it (main.py:1170)
    l2, v2 = net.forward([x[i] for i in range(4)])
    assert np.allclose(l2, logits[:4], atol=1e-5)


@pytest.mark.gpu
def test_facade_default_forward_meets_1e3(tmp_path):
    """policy_value_network(res_block_nums=7).forward — the API north_star names (policy_value_network.py:202-214) with its
    DEFAULT engine (precision "strict": since round 5 k_trunk_mx_c128 up to 8 blocks — fp16 hi halves + the cross terms on a
    block-scaled fp6 MFMA —, k_trunk_split_c128 beyond) — against the fp32 NumPy restatement of the reference graph:
    |dlogit| <= 1e-3 and |dvalue| <= 1e-3 ABSOLUTE on the TF-default weights and on the peaked trained-like set (|logit| ~ 12).
    precision="fp16" selects the fast engine bench.py's headline runs: the same bound on TF-default weights only (measured
    1.2e-4 / 1.1e-4)."""
    from policy_value_network import policy_value_network
    deep = policy_value_network(9, save_dir=str(tmp_path / "deep"))
    assert deep.strict_report()["engine"] == "mx6" and deep.net.mx     # round 6: no depth constant, the measurement decides (TF-default weights: mx6)
    pv = policy_value_network(7, save_dir=str(tmp_path))
    assert pv.net.dtype == torch.float16 and pv.net.backend == "hip" and pv.net.fused_search and pv.net.split and pv.net.mx
    x = _positions(64, 2)
    for wset in ("glorot", "trained_like"):
        H.WEIGHT_SETS[wset](pv.net)
        logits, v = pv.forward(x)
        ln, vn = net_numpy.forward(pv.module.export_tf_layout(), x, 7)
        e = H.errors(logits, v, ln, vn)
        print("facade default (strict = mx6, 7 blocks, %s): max|logit| %.3g dlogit %.3g dvalue %.3g dsoftmax %.3g" % (wset, e["max_abs_logit"], e["dlogit"], e["dvalue"], e["dprob"]))
        assert logits.dtype == np.float32 and logits.shape == (64, 2086) and v.shape == (64, 1)
        assert e["dlogit"] <= 1e-3 and e["dvalue"] <= 1e-3 and e["dprob"] <= 1e-4
    chk = pv.net.range_check()
    assert chk["finite"] and 0 < chk["max_activation"] < 1e4
    fast = policy_value_network(7, save_dir=str(tmp_path), precision="fp16")
    assert fast.net.dtype == torch.float16 and not fast.net.split
    logits, v = fast.forward(x)
    ln, vn = net_numpy.forward(fast.module.export_tf_layout(), x, 7)
    e = H.errors(logits, v, ln, vn)
    assert e["dlogit"] <= 1e-3 and e["dvalue"] <= 1e-3 and e["dprob"] <= 1e-6
    # the package-level error probe bench.py prints (net_error: against the fp32 torch module on the device) agrees
    from cchess_zero_amd.net import net_error
    ne = net_error(fast.net, torch.from_numpy(x).cuda())
    assert abs(ne["dlogit"] - e["dlogit"]) <= 5e-5 and abs(ne["dvalue"] - e["dvalue"]) <= 5e-5


# Fused MFMA net (bf16 / fp16 operands, fp32 accumulate) vs the fp32 NumPy restatement of the reference graph.
# north_star's tolerance (1e-3) is stated for fp32 and is asserted for the fp32 engine above.  A 16-bit tower is a
# different function: every one of the 2*blocks+1 conv layers rounds its activations to 8 (bf16) or 11 (fp16) mantissa
# bits, so its error scales with the logits and grows with depth.  Each entry below is (max |dlogit| / max |logit|,
# max |dsoftmax|, max |dvalue|), set to <= 3x the level MEASURED on an MI355X (tests/measure_net_errors.py ->
# profiles/r02_net_errors.json; 64 corpus positions) — a kernel regression of 3x fails.  Weight sets (nethelpers.py):
# glorot = TF-default initialisation (|logit| ~ 0.1, softmax ~ uniform 4.8e-4); trained_like = positive, peaked policy
# (|logit| ~ 10, top probability 0.2-0.4) and a calibrated value head; structured = tap/channel-asymmetric perturbations.
NET_TOL = {
    # measured (profiles/r02_net_errors.json)     rel 0.00642  dprob 1.19e-07  dvalue 1.14e-04
    ("bf16", 2, "glorot"): (1.6e-2, 3.0e-7, 2.9e-4),
    #                                              rel 0.00789  dprob 4.71e-07  dvalue 8.31e-04
    ("bf16", 7, "glorot"): (2.0e-2, 1.2e-6, 2.1e-3),
    #                                              rel 0.0283   dprob 1.56e-06  dvalue 1.61e-03
    ("bf16", 19, "glorot"): (7.0e-2, 3.9e-6, 4.0e-3),
    #                                              rel 0.00382  dprob 5.68e-06  dvalue 8.0e-03
    ("bf16", 3, "structured"): (9.6e-3, 1.4e-5, 2.0e-2),
    #                                              rel 0.00899  dprob 4.85e-03  dvalue 8.36e-02
    ("bf16", 7, "trained_like"): (2.3e-2, 1.2e-2, 2.1e-1),
    #                                              rel 0.0364   dprob 2.32e-02  dvalue 2.5e-01
    ("bf16", 19, "trained_like"): (9.0e-2, 5.8e-2, 6.0e-1),
    #                                              rel 8.11e-04 dprob 1.5e-08   dvalue 1.88e-05
    ("fp16", 2, "glorot"): (2.0e-3, 3.8e-8, 4.7e-5),
    #                                              rel 9.56e-04 dprob 5.74e-08  dvalue 1.14e-04
    ("fp16", 7, "glorot"): (2.4e-3, 1.4e-7, 2.9e-4),
    #                                              rel 3.66e-03 dprob 2.06e-07  dvalue 2.08e-04
    ("fp16", 19, "glorot"): (9.2e-3, 5.2e-7, 5.2e-4),
    #                                              rel 6.04e-04 dprob 9.1e-07   dvalue 9.62e-04
    ("fp16", 3, "structured"): (1.5e-3, 2.3e-6, 2.4e-3),
    #                                              rel 1.09e-03 dprob 6.6e-04   dvalue 7.2e-03
    ("fp16", 7, "trained_like"): (2.7e-3, 1.65e-3, 1.8e-2),
    #                                              rel 3.51e-03 dprob 2.28e-03  dvalue 2.87e-02
    ("fp16", 19, "trained_like"): (8.8e-3, 5.7e-3, 7.2e-2),
}


@pytest.mark.gpu
@pytest.mark.parametrize("dname,blocks,wset", sorted(NET_TOL))
def test_fused_net_vs_fp32_restatement(dname, blocks, wset):
    from cchess_zero_amd.net import PolicyValueNet
    net = PolicyValueNet(blocks, "cuda:0", {"bf16": torch.bfloat16, "fp16": torch.float16}[dname], seed=1)
    assert net.backend == "hip"
    H.WEIGHT_SETS[wset](net)
    x = _positions(64, 2)
    logits, v = net.forward(x)
    ln, vn = net_numpy.forward(net.module.export_tf_layout(), x, blocks)
    e = H.errors(logits, v, ln, vn)
    tl, tp, tv = NET_TOL[(dname, blocks, wset)]
    print("%s %d-block %s: max|logit| %.3g  dlogit %.3g (rel %.3g, tol %.3g)  dprob %.3g (tol %.3g, max prob %.3g)  dvalue %.3g (tol %.3g)" %
          (dname, blocks, wset, e["max_abs_logit"], e["dlogit"], e["dlogit_rel"], tl, e["dprob"], tp, e["max_prob"], e["dvalue"], tv))
    assert np.isfinite(logits).all() and np.isfinite(v).all()
    assert e["dlogit_rel"] <= tl
    assert e["dprob"] <= tp
    assert e["dvalue"] <= tv
    if dname == "fp16" and wset == "glorot" and blocks <= 7:
        assert e["dlogit"] <= 1e-3 and e["dvalue"] <= 1e-3   # north_star's absolute 1e-3 is met by fp16 up to 7 blocks
    # zero-copy 16-channel planes in the net's own dtype (what cz_search_select writes) give the same bits as repacked f32
    xd = torch.from_numpy(x[:5]).cuda()
    x16 = torch.zeros((5, 9, 10, 16), dtype=net.dtype, device="cuda")
    x16[..., :14] = xd.to(net.dtype)
    l1, v1 = net.forward_device(xd)
    l2, v2 = net.forward_device(x16)
    assert torch.equal(l1, l2) and torch.equal(v1, v2)


@pytest.mark.gpu
@pytest.mark.parametrize("B,residual,relu", [(4, False, True), (7, True, True), (64, True, False), (513, False, False), (1, True, True)])
def test_hip_conv3x3_kernel_vs_torch(B, residual, relu):
    """cz_conv3x3_c128_bf16 (MFMA implicit GEMM, fused bias/residual/ReLU) vs an fp32 torch conv on the
    same bf16-rounded operands.  Tolerance: one bf16 rounding of the output (2^-8 relative) plus the
    bf16 rounding before the residual add; accumulation order differs (fp32)."""
    import torch.nn.functional as F
    from cchess_zero_amd.net import PolicyValueNet
    net = PolicyValueNet(1, "cuda:0", torch.bfloat16, seed=4, backend="hip-layer")
    gen = torch.Generator(device="cuda").manual_seed(B)
    x = torch.randn((B, 90, 128), generator=gen, device="cuda").to(torch.bfloat16)
    # asymmetric, tap- and channel-dependent weights so that any tap/channel/transposition mix-up shows
    w = (torch.randn((128, 128, 3, 3), generator=gen, device="cuda") * 0.05)
    w[:, :, 0, 1] += 0.02
    w[:, :, 2, 0] -= 0.03
    bias = torch.randn(128, generator=gen, device="cuda")
    res = torch.randn((B, 90, 128), generator=gen, device="cuda").to(torch.bfloat16) if residual else None
    wp = w.permute(2, 3, 1, 0).reshape(9, 16, 8, 128).permute(0, 1, 3, 2).contiguous().to(torch.bfloat16)
    out = torch.full((B, 90, 128), float("nan"), device="cuda", dtype=torch.bfloat16)
    net._hip_conv(x, (wp, bias.float().contiguous()), res, out, relu)
    torch.cuda.synchronize()
    xr = x.float().reshape(B, 9, 10, 128).permute(0, 3, 1, 2)
    ref = F.conv2d(xr, w.to(torch.bfloat16).float(), bias, padding=1)
    ref = ref.to(torch.bfloat16).float()  # the kernel rounds to bf16 before the residual add
    if residual:
        ref = ref + res.float().reshape(B, 9, 10, 128).permute(0, 3, 1, 2)
    if relu:
        ref = torch.relu(ref)
    got = out.float().reshape(B, 9, 10, 128).permute(0, 3, 1, 2)
    assert torch.isfinite(got).all()
    err = (got - ref).abs()
    tol = 2.0 ** -7 * ref.abs() + 2e-2
    assert bool((err <= tol).all()), "max err %.4g at %s" % (float(err.max()), tuple(int(i) for i in (err == err.max()).nonzero()[0]))


# blocks -> (max|dlogit| / max|logit|, max|dsoftmax|, max|dvalue|, max|dtrunk| / max|trunk|)
# measured (profiles/r02_net_errors.json hip_vs_torch_bf16/*, GPUTEST log): 2 blocks (8.5e-3, 1.6e-7, 1.9e-4, 7.7e-3); 7 blocks
# (1.1e-2, 6.6e-7, 1.2e-3, 8.2e-3); 19 blocks (3.6e-2, 2.0e-6, 3.4e-3, 2.0e-2)
HIP_VS_TORCH_TOL = {2: (2.2e-2, 4.0e-7, 5.0e-4, 2.0e-2), 7: (2.8e-2, 1.7e-6, 3.0e-3, 2.1e-2), 19: (9.0e-2, 5.0e-6, 8.6e-3, 5.0e-2)}


@pytest.mark.gpu
@pytest.mark.parametrize("backend,blocks,n", [("hip", 7, 37), ("hip-layer", 7, 37), ("hip", 2, 1), ("hip", 19, 130)])
def test_hip_tower_matches_torch_tower(backend, blocks, n):
    """Whole tower: fused single-launch kernel / per-layer kernel vs torch/MIOpen in bf16 (same folded
    weights); odd batch sizes exercise the partial last workgroup."""
    from cchess_zero_amd.net import PolicyValueNet
    a = PolicyValueNet(blocks, "cuda:0", torch.bfloat16, seed=1, backend=backend)
    b = PolicyValueNet(blocks, "cuda:0", torch.bfloat16, seed=1, backend="torch")
    x = torch.from_numpy(_positions(n, 3)).cuda()
    la, va = a.forward_device(x)
    lb, vb = b.forward_device(x)
    pa, pb = torch.softmax(la, 1), torch.softmax(lb, 1)
    print("%s vs torch bf16 tower (%d blocks): max|dlogit| %.4g max|dprob| %.4g max|dv| %.4g" %
          (backend, blocks, float((la - lb).abs().max()), float((pa - pb).abs().max()), float((va - vb).abs().max())))
    # two bf16 evaluations of the same folded weights with different rounding points; tolerances <= 3x the measured
    # level (profiles/r02_net_errors.json: hip_vs_torch_bf16/*), relative to the largest logit / trunk activation
    tl, tp, tv, tt = HIP_VS_TORCH_TOL[blocks]
    assert float((la - lb).abs().max()) <= tl * float(lb.abs().max())
    assert float((pa - pb).abs().max()) <= tp and float((va - vb).abs().max()) <= tv
    # trunk activations themselves (before the heads), elementwise
    ta, tb = a.tower(x).float(), b.tower(x).float()
    assert float((ta - tb).abs().max()) <= tt * float(tb.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 2, 37])
def test_fused_net_kernel_paths_agree(n):
    """cz_net_trunk_bf16 (planes -> head conv outputs, one launch) vs the unfused route
    (torch first conv -> cz_tower_c128_bf16 -> torch head convs) and zero-copy bf16x16 planes vs repacked f32x14."""
    from cchess_zero_amd.net import PolicyValueNet
    net = PolicyValueNet(3, "cuda:0", torch.bfloat16, seed=6, backend="hip")
    gen = torch.Generator().manual_seed(3)
    with torch.no_grad():   # non-trivial first-layer bias/BN so the fused first conv's epilogue is exercised
        net.module.conv_in.conv.bias.copy_((torch.randn(128, generator=gen) * 0.1).cuda())
        net.module.conv_in.moving_var.copy_((torch.rand(128, generator=gen) + 0.5).cuda())
    net.refresh()
    x = torch.from_numpy(_positions(n, 7)).cuda()
    l1, v1 = net.forward_device(x)                       # fused, repacked planes
    x16 = torch.zeros((n, 9, 10, 16), dtype=torch.bfloat16, device="cuda")
    x16[..., :14] = x.to(torch.bfloat16)
    l2, v2 = net.forward_device(x16)                     # fused, zero-copy planes
    assert torch.equal(l1, l2) and torch.equal(v1, v2)
    l3, v3 = net.heads(net.tower(x))                     # unfused route
    assert float((l1 - l3).abs().max()) < 2e-2 * float(l3.abs().max()) + 1e-3
    assert float((torch.softmax(l1, 1) - torch.softmax(l3, 1)).abs().max()) < 1e-3 and float((v1 - v3).abs().max()) < 1e-2


@pytest.mark.gpu
@pytest.mark.parametrize("B", [1, 31, 128, 129, 700])
def test_hip_fc_heads_vs_fp64(B):
    """cz_fc_heads_f32 (policy FC on split-bf16 MFMA, value FCs on the fp32 VALU) vs the same three layers in
    float64 on identical inputs (policy_value_network.py:56-74).  Tolerance: policy logits 2e-4 relative to the
    largest logit (hi*hi + hi*lo + lo*hi keeps ~16 mantissa bits per operand), value 1e-5 absolute."""
    from cchess_zero_amd.net import PolicyValueNet
    net = PolicyValueNet(1, "cuda:0", torch.bfloat16, seed=11, backend="hip")
    gen = torch.Generator().manual_seed(B)
    with torch.no_grad():   # non-zero biases so every term of the layers is exercised
        net.module.policy_fc.bias.copy_((torch.randn(2086, generator=gen) * 0.3).cuda())
        net.module.value_fc1.bias.copy_((torch.randn(256, generator=gen) * 0.3).cuda())
        net.module.value_fc2.bias.copy_((torch.randn(1, generator=gen) * 0.1).cuda())
    net.refresh()
    z = torch.relu(torch.randn((B, 90, 3), generator=gen) * 1.5).cuda()
    logits, value = net._hip_fc_heads(z)
    torch.cuda.synchronize()
    m, zd = net.module, z.double()
    ref_l = zd[:, :, :2].reshape(B, 180) @ m.policy_fc.weight.double().t() + m.policy_fc.bias.double()
    h = torch.relu(zd[:, :, 2] @ m.value_fc1.weight.double().t() + m.value_fc1.bias.double())
    ref_v = torch.tanh(h @ m.value_fc2.weight.double().t() + m.value_fc2.bias.double())
    assert logits.shape == (B, 2086) and value.shape == (B, 1)
    ref_l, ref_v = ref_l.detach(), ref_v.detach()
    dl = float((logits.double() - ref_l).abs().max())
    dv = float((value.double() - ref_v).abs().max())
    print("fc heads B=%d: max|dlogit| %.3g (max|logit| %.3g), max|dv| %.3g" % (B, dl, float(ref_l.abs().max()), dv))
    assert dl < 2e-4 * float(ref_l.abs().max())
    assert dv < 1e-5
    # and against the torch fp32 route the other backends use
    l32, v32 = net.fc_heads(z)
    assert float((logits - l32).abs().max()) < 5e-4 * float(l32.abs().max()) and float((value - v32).abs().max()) < 1e-5


# The STRICT engine (split=True, k_trunk_split_c128: hi + lo halves, three MFMAs per product) is held to north_star's contract
# as written — ABSOLUTE 1e-3 on raw logits and on the value against the fp32 restatement of the reference graph — on every
# weight set, including the peaked trained-like one (|logit| ~ 10) and 19 blocks, where a 16-bit tower is 15x .. 50x off.
# Emulated on the CPU (tools/precision_decomposition.py; against the fp64 graph): fp16 halves 1.8e-5 / 1.3e-5 at 7 blocks and
# 4e-5 / 1e-4 at 19 (trained-like), bf16 halves 1.8e-4 / 1.3e-4 and 4.8e-4 / 4.1e-4.
STRICT_CASES = [("fp16", 2, "glorot"), ("fp16", 7, "glorot"), ("fp16", 3, "structured"), ("fp16", 7, "trained_like"),
                ("fp16", 19, "glorot"), ("fp16", 19, "trained_like"), ("bf16", 7, "trained_like"), ("bf16", 19, "trained_like")]


@pytest.mark.gpu
@pytest.mark.parametrize("dname,blocks,wset", STRICT_CASES)
def test_strict_engine_meets_1e3_absolute(dname, blocks, wset):
    from cchess_zero_amd.net import PolicyValueNet
    net = PolicyValueNet(blocks, "cuda:0", {"bf16": torch.bfloat16, "fp16": torch.float16}[dname], seed=1, split=True)
    assert net.backend == "hip" and net.split and net.fused_search
    H.WEIGHT_SETS[wset](net)
    x = _positions(64, 2)
    logits, v = net.forward(x)
    ln, vn = net_numpy.forward(net.module.export_tf_layout(), x, blocks)
    e = H.errors(logits, v, ln, vn)
    print("strict %sx2 %d-block %s: max|logit| %.3g  dlogit %.3g (rel %.3g)  dprob %.3g  dvalue %.3g  argmax agreement %.3f" %
          (dname, blocks, wset, e["max_abs_logit"], e["dlogit"], e["dlogit_rel"], e["dprob"], e["dvalue"], e["argmax_agree"]))
    assert np.isfinite(logits).all() and np.isfinite(v).all()
    assert e["dlogit"] <= 1e-3 and e["dvalue"] <= 1e-3     # north_star: "policy/value outputs match within 1e-3 fp32"
    if dname == "fp16":                                   # and the fp16-halves engine with a decade to spare
        assert e["dlogit"] <= 2e-4 and e["dvalue"] <= 2.5e-4
    assert e["argmax_agree"] == 1.0


# The MX engine (split="mx", k_trunk_mx_c128; round 5): a*w = a_hi*w_hi on fp16 MFMAs + both cross terms of the hi + lo split on
# one block-scaled fp6 MFMA (4 significant bits per cross-term operand, ~2^-16 of a product).  Held to north_star's ABSOLUTE
# 1e-3 at the depths precision "strict" uses it for (<= 8 blocks) — emulated on the CPU (tests/mxemu.py; against the fp64
# graph): 4.9e-4 / 2.2e-4 on trained-like weights at 7 blocks, 3.2e-4 / 1.9e-4 at 8; 1.0e-3 / 1.1e-3 at 19 (k_trunk_split_c128's
# job) — and, at shallow depth, to its CPU emulation on the trunk activations themselves.
MX_CASES = [(2, "glorot"), (7, "glorot"), (3, "structured"), (3, "trained_like"), (7, "trained_like"), (8, "trained_like")]


@pytest.mark.gpu
@pytest.mark.parametrize("blocks,wset", MX_CASES)
def test_mx_engine_meets_1e3_absolute(blocks, wset):
    from cchess_zero_amd.net import PolicyValueNet
    net = PolicyValueNet(blocks, "cuda:0", torch.float16, seed=1, split="mx")
    assert net.backend == "hip" and net.split and net.mx and net.fused_search
    H.WEIGHT_SETS[wset](net)
    x = _positions(64, 2)
    logits, v = net.forward(x)
    ln, vn = net_numpy.forward(net.module.export_tf_layout(), x, blocks)
    e = H.errors(logits, v, ln, vn)
    print("mx6 %d-block %s: max|logit| %.3g  dlogit %.3g (rel %.3g)  dprob %.3g  dvalue %.3g  argmax agreement %.3f" %
          (blocks, wset, e["max_abs_logit"], e["dlogit"], e["dlogit_rel"], e["dprob"], e["dvalue"], e["argmax_agree"]))
    assert np.isfinite(logits).all() and np.isfinite(v).all()
    assert e["dlogit"] <= 1e-3 and e["dvalue"] <= 1e-3     # north_star: "policy/value outputs match within 1e-3 fp32"
    if wset != "trained_like":
        assert e["dlogit"] <= 5e-5 and e["dvalue"] <= 5e-5
    assert e["argmax_agree"] == 1.0


@pytest.mark.gpu
def test_mx_engine_at_19_blocks_is_at_the_edge_and_strict_measures_it():
    """At 19 blocks k_trunk_mx_c128's error on the peaked trained-like set of this suite sits AT north_star's 1e-3 (measured on the
    MI355X 1.02e-3 / 7.2e-4, emulated 1.02e-3 / 1.14e-3; on TF-default weights 1.5e-5).  Round 5 kept it away from deep nets with a
    depth constant; since round 6 precision "strict" measures: on these weights it falls over to the three-MFMA engine (4.1e-5 /
    1.8e-4), on TF-default weights of the same depth it stays on mx6."""
    from cchess_zero_amd.net import STRICT_CHECK_TOL, PolicyValueNet
    net = PolicyValueNet(19, "cuda:0", torch.float16, seed=1, split="mx")
    x = _positions(64, 2)
    auto0 = PolicyValueNet(19, "cuda:0", torch.float16, split="strict", module=net.module)       # TF-default weights
    rep0 = auto0.strict_check()
    assert rep0["engine"] == "mx6" and not rep0["fell_over_from"] and max(rep0["dlogit"], rep0["dvalue"]) <= 1e-4
    H.trained_like_(net)
    logits, v = net.forward(x)
    ln, vn = net_numpy.forward(net.module.export_tf_layout(), x, 19)
    e = H.errors(logits, v, ln, vn)
    print("mx6 19-block trained_like (explicit choice): dlogit %.3g dvalue %.3g argmax agreement %.3f" % (e["dlogit"], e["dvalue"], e["argmax_agree"]))
    assert e["dlogit"] <= 2e-3 and e["dvalue"] <= 2e-3 and e["argmax_agree"] == 1.0
    auto = PolicyValueNet(19, "cuda:0", torch.float16, split="strict", module=net.module)
    l2, v2 = auto.forward(x)                             # the pending measurement happens here
    rep = auto.strict_report
    print("strict on the 19-block trained-like set:", rep)
    assert rep["engine"] == "fp16x2" and rep["fell_over_from"][0]["engine"] == "mx6" and auto.split and not auto.mx
    assert max(rep["fell_over_from"][0]["dlogit"], rep["fell_over_from"][0]["dvalue"]) > STRICT_CHECK_TOL
    e2 = H.errors(l2, v2, ln, vn)
    assert e2["dlogit"] <= 2e-4 and e2["dvalue"] <= 2.5e-4


@pytest.mark.gpu
@pytest.mark.parametrize("blocks,wset,tol", [(1, "structured", 2e-5), (1, "trained_like", 2e-5), (2, "structured", 4e-5)])
def test_mx_engine_matches_its_cpu_emulation(blocks, wset, tol):
    """k_trunk_mx_c128's trunk activations (fp32, last layer) against tests/mxemu.py — the same hi / lo split, the same
    E2M3 grid under the same block scales, fp32 accumulation in another order: measured 3e-6 .. 6e-6 of the largest
    activation at 1 - 2 blocks (a wrong slot, scale byte or channel grouping is off by 1e-3 and more); and every head output."""
    import mxemu
    from cchess_zero_amd.net import PolicyValueNet
    net = PolicyValueNet(blocks, "cuda:0", torch.float16, seed=1, split="mx")
    H.WEIGHT_SETS[wset](net)
    x = _positions(37, 2)
    xd = torch.from_numpy(x).cuda()
    lg, vg = net.forward_device(xd)
    tg = net.tower(xd).float().cpu()
    mod = net.module.float().cpu()
    with torch.no_grad():
        le, ve, te = mxemu.forward_mx(mod, torch.from_numpy(x).permute(0, 3, 1, 2).contiguous())
    net.module.to("cuda:0")
    rel = float((tg - te).abs().max() / te.abs().max())
    dl, dv = float((lg.cpu() - le).abs().max()), float((vg.cpu().reshape(-1) - ve.reshape(-1)).abs().max())
    print("mx6 %d-block %s kernel vs CPU emulation: trunk %.3g of its largest value, logits %.3g (max |logit| %.3g), value %.3g" %
          (blocks, wset, rel, dl, float(le.abs().max()), dv))
    assert rel <= tol
    assert dl <= 2e-5 * float(le.abs().max()) + 2e-6 and dv <= 5e-5


@pytest.mark.gpu
@pytest.mark.parametrize("dt,split", [(torch.float16, True), (torch.bfloat16, True), (torch.float16, "mx")])
def test_strict_engine_rows_are_independent_and_routes_agree(dt, split):
    """k_trunk_split_c128 / k_trunk_mx_c128: (i) a position's outputs do not depend on its row in the batch, on the batch size (ragged last
    workgroup: 2 positions per workgroup) or on the device-side row count of the compact path; (ii) zero-copy 16-channel planes
    in the operand type == repacked f32 planes; (iii) the fp32 trunk output (hi + lo) through torch's fp32 head convs and FCs
    agrees with the fused heads to fp32 summation-order noise."""
    import ctypes as C
    from cchess_zero_amd._lib import check, lib
    from cchess_zero_amd.net import PolicyValueNet
    net = PolicyValueNet(3, "cuda:0", dt, seed=4, split=split)
    H.structured_(net)
    x = torch.from_numpy(_positions(37, 5)).cuda()
    l_all, v_all = net.forward_device(x)
    for idx in ([0], [36], [5, 0, 36], list(range(36, -1, -1))):
        li, vi = net.forward_device(x[idx])
        assert torch.equal(li, l_all[idx]) and torch.equal(vi, v_all[idx]), idx
    x16 = torch.zeros((37, 9, 10, 16), dtype=dt, device="cuda")
    x16[..., :14] = x.to(dt)
    l2, v2 = net.forward_device(x16)
    assert torch.equal(l2, l_all) and torch.equal(v2, v_all)
    # compact path: only the first *n_rows rows are computed, and they are the same bits
    n = torch.tensor([11], dtype=torch.int32, device="cuda")
    z_full = net._hip_net_forward(x16)
    check(lib().cz_set_batch_count(net._hip_ctx().h, C.c_void_p(n.data_ptr())), "cz_set_batch_count")
    try:
        z_part = torch.full_like(z_full, float("nan"))
        z_tmp = net._hip_net_forward(x16)
        z_part[:11] = z_tmp[:11]
    finally:
        check(lib().cz_set_batch_count(net._hip_ctx().h, None), "cz_set_batch_count")
    assert torch.equal(z_part[:11], z_full[:11])
    # trunk route
    l3, v3 = net.heads(net.tower(x))
    dl, dv = float((l_all - l3).abs().max()), float((v_all - v3).abs().max())
    print("strict %s trunk route vs fused heads: max|dlogit| %.3g (max|logit| %.3g) max|dvalue| %.3g" % (dt, dl, float(l3.abs().max()), dv))
    assert dl <= 2e-5 * float(l3.abs().max()) + 1e-6 and dv <= 2e-6


@pytest.mark.gpu
def test_fp16_engines_do_not_overflow_to_inf():
    """a checkpoint whose activations exceed 65504 must not put inf / NaN into the priors.  The fp16 kernels clamp
    at the largest finite half on the store (k_tower8_c128: packed min after the packed max; k_trunk_split_c128: v_med3 before
    the split); with the first layer's bias at 1e5 every activation of the net saturates and the outputs stay finite."""
    from cchess_zero_amd.net import PolicyValueNet
    x = torch.from_numpy(_positions(6, 3)).cuda()
    for split in (False, True, "mx"):
        net = PolicyValueNet(2, "cuda:0", torch.float16, seed=2, split=split)
        with torch.no_grad():
            net.module.conv_in.conv.bias.fill_(1.0e5)
        net.refresh()
        logits, v = net.forward_device(x)
        assert bool(torch.isfinite(logits).all()) and bool(torch.isfinite(v).all()), split


@pytest.mark.gpu
@pytest.mark.parametrize("blocks,n", [(2, 5), (7, 33)])
def test_fp16_trunk_route_consistent(blocks, n):
    """cz_net_trunk_f16: the trunk-output route (tower() + torch heads) agrees with the fused heads route."""
    from cchess_zero_amd.net import PolicyValueNet
    net = PolicyValueNet(blocks, "cuda:0", torch.float16, seed=3)
    assert net.backend == "hip"
    xd = torch.from_numpy(_positions(n, 11)).cuda()
    l1, v1 = net.forward_device(xd)
    l3, v3 = net.heads(net.tower(xd))
    dl, dv = float((l1 - l3).abs().max()), float((v1 - v3).abs().max())
    print("fp16 %d-block trunk route vs fused heads: max|dlogit| %.3g (max|logit| %.3g) max|dvalue| %.3g" % (blocks, dl, float(l3.abs().max()), dv))
    # measured: 5.5e-6 of the largest logit, 2e-7 on the value (fp32 summation order of the head convs only)
    assert dl <= 2e-5 * float(l3.abs().max()) and dv <= 1e-6


def _heavy_tailed_(net, logit_scale=80.0, seed=17):
    """A weight set that is harder on a max-scaled 16-channel E2M3 block than nethelpers.trained_like_: the
    trained-like set, then heavy-tailed BN variances (log-normal, sigma 0.7: under quirk Q5 a trained checkpoint's activations
    are unnormalised, a few channels dominate their block) and a policy FC rescaled to |logit| ~ logit_scale."""
    H.trained_like_(net)
    gen = torch.Generator().manual_seed(seed)
    m = net.module
    with torch.no_grad():
        for cb in m.convbns()[:-2]:
            cb.moving_var.copy_(torch.exp(torch.randn(cb.moving_var.shape, generator=gen) * 0.7).to(cb.moving_var.device))
        x = torch.from_numpy(_positions(96, 123)).to(m.policy_fc.weight.device).permute(0, 3, 1, 2)
        logits, _ = m(x)
        m.policy_fc.weight.mul_(logit_scale / float(logits.max(dim=1).values.mean()))
    net.refresh()
    return net


@pytest.mark.gpu
def test_strict_is_measured_and_falls_over_when_mx6_misses(tmp_path):
    """precision "strict" is a guarantee.  On TF-default weights the facade's 7-block net measures itself
    (64 distinct positions, engine vs fp32 module) and stays on k_trunk_mx_c128; on a weight set that breaks mx6 at 7 blocks
    (heavy-tailed BN variances, |logit| ~ 80: mx6's ~1e-5 of the largest logit — measured on the MI355X 2.9e-4 at |logit| 32,
    emulated 6.3e-4 / 1.3e-3 at 35 / 70 on the corpus positions — is above the 5e-4 it allows itself) the SAME facade object falls
    over to k_trunk_split_c128, reports what it measured, and forward() still meets north_star's 1e-3 against the NumPy
    restatement of the reference graph (policy_value_network.py:202-214); after benign weights are restored it is back on mx6."""
    import warnings
    from cchess_zero_amd.net import STRICT_CHECK_POSITIONS, STRICT_CHECK_TOL, PolicyValueNet
    from policy_value_network import policy_value_network
    pv = policy_value_network(7, save_dir=str(tmp_path))
    rep = pv.strict_report()
    assert rep["engine"] == "mx6" and rep["positions"] == STRICT_CHECK_POSITIONS >= 64 and not rep["fell_over_from"]
    assert rep["dlogit"] <= 5e-5 and rep["dvalue"] <= 5e-5 and pv.net.mx and pv.net.fused_search
    benign = {k: v.clone() for k, v in pv.module.state_dict().items()}
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        _heavy_tailed_(pv.net)
        x = _positions(64, 2)
        logits, v = pv.forward(x)                       # the pending measurement happens here
    rep = pv.strict_report()
    print("strict on heavy-tailed weights:", rep)
    assert rep["fell_over_from"] and rep["fell_over_from"][0]["engine"] == "mx6"
    assert max(rep["fell_over_from"][0]["dlogit"], rep["fell_over_from"][0]["dvalue"]) > STRICT_CHECK_TOL
    assert rep["engine"] in ("fp16x2", "fp32") and max(rep["dlogit"], rep["dvalue"]) <= STRICT_CHECK_TOL
    assert any("falling over" in str(i.message) for i in w)
    assert not pv.net.mx and pv.net.engine_name == rep["engine"]
    ln, vn = net_numpy.forward(pv.module.export_tf_layout(), x, 7)
    e = H.errors(logits, v, ln, vn)
    print("facade after fall-over: max|logit| %.3g dlogit %.3g dvalue %.3g" % (e["max_abs_logit"], e["dlogit"], e["dvalue"]))
    assert e["max_abs_logit"] > 50 and e["dlogit"] <= 1e-3 and e["dvalue"] <= 1e-3 and e["argmax_agree"] == 1.0
    # the engine it fell over FROM, run by explicit choice on the same weights, does miss the contract on these inputs
    mx = PolicyValueNet(7, "cuda:0", torch.float16, split="mx", module=pv.module)
    lm, vm = mx.forward(x)
    em = H.errors(lm, vm, ln, vn)
    print("mx6 by explicit choice on the same weights: dlogit %.3g dvalue %.3g" % (em["dlogit"], em["dvalue"]))
    assert max(em["dlogit"], em["dvalue"]) > STRICT_CHECK_TOL
    # benign weights again (what a train_step / restore does: refresh() restarts the ladder)
    pv.module.load_state_dict(benign)
    pv.refresh()
    assert pv.strict_report()["engine"] == "mx6" and pv.net.mx


@pytest.mark.gpu
def test_strict_last_rung_fp32_drives_the_fused_search_loop():
    """The ladder's last rung (torch / MIOpen fp32) under a SearchEngine that was built for the fused engine (fp16 planes, 16
    channels): forced by tol = 0, the lock-step loop switches to the full-logits expansion and finds the visit counts of a
    plain fp32 engine."""
    from cchess_zero_amd.engine import SearchEngine
    from cchess_zero_amd.net import PolicyValueNet
    from cchess_zero_amd.rules import START_BOARD
    G = 32
    net = PolicyValueNet(2, "cuda:0", torch.float16, seed=3, split="strict")
    rep = net.strict_check(tol=0.0)
    assert rep["engine"] == "fp32" and [f["engine"] for f in rep["fell_over_from"]] == ["mx6", "fp16x2"] and not net.fused_search
    assert rep["dlogit"] <= 1e-5 and rep["dvalue"] <= 1e-5
    ref = PolicyValueNet(2, "cuda:0", torch.float32, module=net.module, backend="torch")
    boards, side = np.tile(START_BOARD, (G, 1)), np.zeros(G, np.uint8)
    ea = SearchEngine(G, 4096, plane_dtype=torch.float16, channels=16)
    eb = SearchEngine(G, 4096, plane_dtype=torch.float32, channels=14)
    for e, n in ((ea, net), (eb, ref)):
        e.reset(boards, side, None)
        e.search(n.forward_device, 24)
    a, b = ea.root_stats_host(), eb.root_stats_host()
    assert np.array_equal(a["N"], b["N"]) and np.array_equal(a["label"], b["label"])


def test_strict_ladder_logic_cpu():
    """The control flow of precision "strict" without a GPU: a stand-in whose engines deviate from the fp32 module by a chosen
    amount per rung.  The ladder goes mx6 -> fp16x2 -> fp32 exactly as far as the measurement demands, reports what it measured
    and what it fell over from, treats a non-finite engine as a failure, starts over after refresh(), and never leaves the last
    rung."""
    import warnings
    from cchess_zero_amd.net import STRICT_CHECK_TOL, PolicyValueModule, PolicyValueNet

    class Fake(PolicyValueNet):
        def __init__(self, err_by_rung):
            self.device, self.dtype = torch.device("cpu"), torch.float16
            self.module = PolicyValueModule(1, seed=2)
            self.res_block_nums = 1
            self.strict_auto, self.strict_report, self._check_pending, self._fp32_fallback = True, None, False, False
            self.err_by_rung, self.packs = err_by_rung, []
            self.refresh()

        def _pack(self):
            self.packs.append(self._rung)

        @torch.no_grad()
        def forward_device(self, planes):
            lg, v = self.module(planes.permute(0, 3, 1, 2).contiguous())
            e = self.err_by_rung[self._rung]
            return lg + e, v + (0.5 * e if e == e else e)

    x = torch.from_numpy(_positions(8, 4))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        n = Fake({0: 1e-5, 1: 0.0, 2: 0.0})
        r = n.strict_check(x)
        assert r["engine"] == "mx6" and not r["fell_over_from"] and abs(r["dlogit"] - 1e-5) < 2e-6 and r["positions"] == 8 and r["tol"] == STRICT_CHECK_TOL
        assert n.mx and n.split and n.backend == "hip" and not w
        n = Fake({0: 8e-4, 1: 2e-5, 2: 0.0})
        r = n.strict_check(x)
        assert r["engine"] == "fp16x2" and [f["engine"] for f in r["fell_over_from"]] == ["mx6"] and abs(r["fell_over_from"][0]["dlogit"] - 8e-4) < 2e-5
        assert not n.mx and n.split and n.backend == "hip" and n.packs == [0, 1] and len(w) == 1 and "falling over to fp16x2" in str(w[0].message)
        n.refresh()                                   # new weights: the ladder starts over, the measurement is pending
        assert n.mx and n._rung == 0 and n._check_pending is False      # (pending only with a GPU: this stand-in lives on the CPU)
        n = Fake({0: float("nan"), 1: 6e-4, 2: 1e-7})
        r = n.strict_check(x)
        assert r["engine"] == "fp32" and [f["engine"] for f in r["fell_over_from"]] == ["mx6", "fp16x2"] and n.backend == "torch" and n._fp32_fallback
        assert r["fell_over_from"][0]["dlogit"] != r["fell_over_from"][0]["dlogit"]       # NaN recorded, counted as a failure
        n = Fake({0: 1.0, 1: 1.0, 2: 1.0})          # nothing meets the tolerance: the last rung is where it stays, and says so
        r = n.strict_check(x)
        assert r["engine"] == "fp32" and r["dlogit"] > STRICT_CHECK_TOL and len(r["fell_over_from"]) == 2
        assert n.strict_check(x, tol=2.0)["engine"] == "fp32"           # a later check does not climb back up by itself (refresh does)
