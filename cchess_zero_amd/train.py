"""Training step and policy update of cchess-zero on PyTorch-ROCm, data-parallel over ranks.

  reference                                                       here
  policy_value_network.py:77-92   loss = MSE(z, v) + CE(pi, logits) + 1e-4 * sum(w^2)/2 over ALL trainables   Trainer.loss
  :95-96,117-126  MomentumOptimizer(momentum 0.9, use_nesterov), clip_by_global_norm(100), check_numerics        Trainer.train_step
  :186-199        train_step(positions, probs, winners, lr) -> (accuracy, loss, global_step)                     Trainer.train_step
  main.py:1157-1204  policy_update: 5 epochs on one mini-batch, KL early stop, lr_multiplier x / 1.5             policy_update()
  policy_value_network_gpus.py:216-250  average_gradients over in-process towers                                 one flat all-reduce

Everything here is plain torch and runs on whatever device the module lives on (the GPU in the product, the CPU in the
gloo world-2 test).  With torch.distributed initialised every rank must make the SAME control-flow decisions, or the
collectives of different ranks stop matching: the mini-batch is drawn with a generator seeded from the global step (all
ranks hold the same gathered buffer), each rank trains on its slice of it, gradients are averaged, and the KL that drives
the early stop and the learning-rate adaptation is all-reduced before it is looked at.
"""
import contextlib
import os
import random
import time

import numpy as np
import torch
import torch.distributed as dist
import torch.nn.functional as F

from .parallel import allreduce_gradients, broadcast_weights


def _dist_on():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


@contextlib.contextmanager
def deterministic_convolutions(on=True):
    """Restrict MIOpen to its deterministic convolution kernels for the calls made inside (torch hands the flag to MIOpen as
    MIOPEN_CONVOLUTION_ATTRIB_DETERMINISTIC at every convolution call, forward and backward).  Measured on MI355X, round 3
    (tools/train_step_time.py, tools/train_algo_probe.sh, tests/test_train.py): left to itself MIOpen runs this net's 3x3
    layers as fp32 implicit-GEMM kernels whose weight-gradient variant reduces split-K partial sums with atomics
    (igemm_wrw_gtcx35_nhwc_fp32_*_gkgs) — 15 ms per 7-block step at batch 512, results that differ in the last bits from run
    to run; restricted, it falls back to its naive kernels with float64 accumulation (naive_conv_ab_nonpacked_*_float_double_
    float): bit-identical from run to run and the closest to the float64 restatement, but 1.2 s per step.  Hence opt-in
    (Trainer(deterministic=True) or CCHESS_TRAIN_DETERMINISTIC=1): for debugging and for the parity test, not for training."""
    if not on:
        yield
        return
    prev = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    try:
        yield
    finally:
        torch.backends.cudnn.deterministic = prev


class Trainer:
    """One optimiser over a PolicyValueModule.  Not tied to a device: tensors follow the module's parameters."""

    def __init__(self, module, c_l2=0.0001, momentum=0.9, global_norm=100.0, deterministic=None):
        self.module = module
        self.deterministic = (os.environ.get("CCHESS_TRAIN_DETERMINISTIC", "0") == "1") if deterministic is None else bool(deterministic)
        self.c_l2, self.momentum, self.global_norm = float(c_l2), float(momentum), float(global_norm)
        self.opt = torch.optim.SGD(module.parameters(), lr=0.0, momentum=self.momentum, nesterov=True)
        self.global_step = 0
        self.last_grad_norm = float("nan")

    @property
    def device(self):
        return next(self.module.parameters()).device

    def _tensors(self, positions, probs, winners):
        dev = self.device
        x = torch.as_tensor(np.asarray(positions, dtype=np.float32)).to(dev).permute(0, 3, 1, 2)
        pi = torch.as_tensor(np.asarray(probs, dtype=np.float32)).to(dev)
        z = torch.as_tensor(np.asarray(winners, dtype=np.float32)).to(dev).reshape(-1, 1)
        return x, pi, z

    def loss(self, positions, probs, winners, training=True):
        """-> (loss, accuracy): policy_value_network.py:77-92,108-111.  training=True uses batch statistics in the
        BatchNorm layers (is_training=True; the moving averages are never updated, quirk Q5)."""
        x, pi, z = self._tensors(positions, probs, winners)
        logits, v = self.module(x, training=training)
        policy_loss = -(pi * F.log_softmax(logits, dim=1)).sum(dim=1).mean()   # softmax_cross_entropy_with_logits
        value_loss = F.mse_loss(v, z)                                          # tf.losses.mean_squared_error
        l2 = sum((p * p).sum() for p in self.module.parameters()) * (self.c_l2 / 2.0)  # l2_regularizer over ALL trainables
        accuracy = (logits.argmax(dim=1) == pi.argmax(dim=1)).float().mean()
        return value_loss + policy_loss + l2, accuracy

    def train_step(self, positions, probs, winners, learning_rate):
        """-> (accuracy, loss, global_step), like policy_value_network.py:186-199.  Under torch.distributed the gradients
        are averaged over the ranks before clipping (every rank then applies the identical update)."""
        for g in self.opt.param_groups:
            g["lr"] = float(learning_rate)
        self.module.train()
        self.opt.zero_grad(set_to_none=True)
        with deterministic_convolutions(self.deterministic):
            loss, accuracy = self.loss(positions, probs, winners, training=True)
            loss.backward()
        allreduce_gradients(self.module)                                       # no-op without a process group
        grads = [p.grad for p in self.module.parameters() if p.grad is not None]
        # tf.clip_by_global_norm: t * clip_norm / max(global_norm, clip_norm)
        gn = torch.sqrt(sum((g.double() * g.double()).sum() for g in grads))
        scale = self.global_norm / max(float(gn), self.global_norm)
        for g in grads:                                                        # tf.check_numerics('NaN Found!')
            if scale != 1.0:
                g.mul_(scale)
            if not torch.isfinite(g).all():
                raise FloatingPointError("NaN Found!")
        self.last_grad_norm = float(gn)
        self.opt.step()
        self.module.eval()
        self.global_step += 1
        if _dist_on():   # the loss / accuracy reported are the means over the ranks' slices
            t = torch.stack([accuracy.detach().double(), loss.detach().double()])
            dist.all_reduce(t)
            t /= dist.get_world_size()
            return float(t[0]), float(t[1]), self.global_step
        return float(accuracy.detach()), float(loss.detach()), self.global_step

    def load_tf_momentum(self, slots):
        """Momentum slots in the reference's TF layout (net.momentum_slots_from_tf_variables: HWIO kernels, [in,out] FC
        weights) -> this optimiser's momentum buffers.  tf.train.MomentumOptimizer's accumulator (accum = momentum * accum +
        grad) is the quantity torch.optim.SGD keeps as momentum_buffer, so training resumes where the checkpoint stopped."""
        m = self.module
        pairs = []
        for i, cb in enumerate(m.convbns()):
            pairs.append((cb.conv.weight, torch.from_numpy(np.asarray(slots["conv%d/kernel" % i], np.float32)).permute(3, 2, 0, 1)))
            pairs.append((cb.conv.bias, torch.from_numpy(np.asarray(slots["conv%d/bias" % i], np.float32))))
        for name, fc in (("policy_fc", m.policy_fc), ("value_fc1", m.value_fc1), ("value_fc2", m.value_fc2)):
            pairs.append((fc.weight, torch.from_numpy(np.asarray(slots[name + "/weights"], np.float32)).t()))
            pairs.append((fc.bias, torch.from_numpy(np.asarray(slots[name + "/biases"], np.float32))))
        for p, buf in pairs:
            if tuple(buf.shape) != tuple(p.shape):
                raise ValueError("momentum slot of shape %s for a parameter of shape %s" % (tuple(buf.shape), tuple(p.shape)))
            self.opt.state[p]["momentum_buffer"] = buf.contiguous().to(p.device, p.dtype).clone()

    def tf_momentum_slots(self):
        """The inverse of load_tf_momentum: this optimiser's momentum buffers in the reference's TF layout, keyed like
        export_tf_layout() (zeros for parameters that have not been stepped yet — a fresh MomentumOptimizer slot)."""
        m = self.module
        out = {}

        def buf(p):
            st = self.opt.state.get(p, {})
            b = st.get("momentum_buffer")
            return (b if b is not None else torch.zeros_like(p)).detach().float().cpu()
        for i, cb in enumerate(m.convbns()):
            out["conv%d/kernel" % i] = buf(cb.conv.weight).permute(2, 3, 1, 0).contiguous().numpy().copy()
            out["conv%d/bias" % i] = buf(cb.conv.bias).numpy().copy()
        for name, fc in (("policy_fc", m.policy_fc), ("value_fc1", m.value_fc1), ("value_fc2", m.value_fc2)):
            out[name + "/weights"] = buf(fc.weight).t().contiguous().numpy().copy()
            out[name + "/biases"] = buf(fc.bias).numpy().copy()
        return out

    def state_dict(self):
        """Model, momentum buffers (tf.train.Saver persists the Momentum slot variables) and the step."""
        return {"model": self.module.state_dict(), "optimizer": self.opt.state_dict(), "global_step": int(self.global_step)}

    def load_state_dict(self, d):
        self.module.load_state_dict(d["model"])
        if d.get("optimizer") is not None:
            self.opt.load_state_dict(d["optimizer"])
        self.global_step = int(d.get("global_step", 0))


def kl_estimate_rows(old_probs, new_probs):
    """The reference's per-row KL estimate on what forward() returns — RAW LOGITS (main.py:1175-1181): the terms
    old * log((old + 1e-10) / (new + 1e-10)), minus the ones that print as 'nan' or 'inf' (a '-inf' term is kept, like
    there), summed per row — in the dtype forward() returns (float32), as np.sum / np.mean do there."""
    with np.errstate(all="ignore"):
        kl_tmp = old_probs * (np.log((old_probs + 1e-10) / (new_probs + 1e-10)))
    return np.array([np.sum(line[~(np.isnan(line) | (line == np.inf))]) for line in kl_tmp])


def policy_update(net, data_buffer, batch_size, epochs, learning_rate, lr_multiplier, kl_targ, seed=0, log=print, temperature=1.0,
                  save=True, sample=None):
    """cchess_main.policy_update (main.py:1157-1204).  net: forward(list of planes) -> (logits, value) ndarrays,
    train_step(...) -> (accuracy, loss, global_step), save(step), global_step.  Returns (lr_multiplier, info dict).

    data_buffer items: dense (planes, pi, z) tuples like the reference's, or packed 608-byte self-play records.

    Rank-consistent by construction: the mini-batch indices come from random.Random(seed, global step) — identical on
    every rank because every rank holds the same gathered buffer —, rank r trains on elements r::world of it, and the
    KL estimate is averaged over the ranks before the early-stop / learning-rate decisions.
    save=False skips the checkpoint (the batched loop of main.py saves once per self-play batch, not once per update);
    sample: explicit buffer indices instead of the seeded draw (tests)."""
    world = dist.get_world_size() if _dist_on() else 1
    rank = dist.get_rank() if _dist_on() else 0
    rng = random.Random((int(seed) << 32) ^ int(net.global_step))
    if batch_size < world:
        raise ValueError("batch_size %d is smaller than the number of ranks %d: a rank would train on nothing" % (batch_size, world))
    # main.py:1159 (random.sample): indices are drawn, the buffer itself (possibly hundreds of thousands of records) is not copied
    idx = list(sample) if sample is not None else rng.sample(range(len(data_buffer)), batch_size)
    mini_batch = [data_buffer[i] for i in idx[rank::world]]
    if isinstance(mini_batch[0], np.ndarray) and mini_batch[0].dtype == np.uint8 and mini_batch[0].ndim == 1:
        # the buffer holds PACKED records (608 bytes each instead of 22 KB of dense planes + pi): only the mini-batch is
        # expanded to the reference's (state planes, pi[2086], z) tuples, pi with the reference's exact float64 expression
        from .selfplay import to_dense
        planes, pi, z = to_dense(np.stack(mini_batch), temperature, exact=True)
        state_batch, mcts_probs_batch, winner_batch = list(planes), list(pi), np.expand_dims(z, 1)
    else:
        state_batch = [d[0] for d in mini_batch]
        mcts_probs_batch = [d[1] for d in mini_batch]
        winner_batch = np.expand_dims([d[2] for d in mini_batch], 1)
    start_time = time.time()
    old_probs, old_v = net.forward(state_batch)
    kl, loss, accuracy, new_v, steps = 0.0, 0.0, 0.0, old_v, 0
    for i in range(epochs):
        accuracy, loss, global_step = net.train_step(state_batch, mcts_probs_batch, winner_batch, learning_rate * lr_multiplier)
        steps += 1
        new_probs, new_v = net.forward(state_batch)
        kl_rows = kl_estimate_rows(old_probs, new_probs)   # the reference feeds raw logits into its KL estimate (main.py:1175)
        if world > 1:
            t = torch.tensor([float(kl_rows.astype(np.float64).sum()), float(len(kl_rows))], dtype=torch.float64)
            if dist.get_backend() == "nccl":
                t = t.cuda()
            dist.all_reduce(t)
            kl = float(t[0] / t[1])
        else:
            kl = float(np.mean(kl_rows))
        if kl > kl_targ * 4:   # early stopping if D_KL diverges badly — the same decision on every rank
            break
    if world > 1:
        broadcast_weights(net.module, src=0)   # replicas stay bit-identical whatever the reduction order did
        if hasattr(net, "refresh"):
            net.refresh()
    if rank == 0 and save:
        net.save(net.global_step)
    log("train using time {} s".format(time.time() - start_time))
    if kl > kl_targ * 2 and lr_multiplier > 0.1:
        lr_multiplier /= 1.5
    elif kl < kl_targ / 2 and lr_multiplier < 10:
        lr_multiplier *= 1.5
    # the figures the reference logs (main.py:1197-1198): winner_batch is [B,1] there and old_v.flatten() is [B], so the
    # difference broadcasts to a [B,B] matrix of z_i - v_j — reproduced as written (a report, nothing is decided on it)
    wb = np.array(winner_batch)
    with np.errstate(all="ignore"):
        info = dict(kl=kl, lr_multiplier=lr_multiplier, loss=loss, accuracy=accuracy, steps=steps,
                    explained_var_old=1 - np.var(wb - np.asarray(old_v).flatten()) / np.var(wb),
                    explained_var_new=1 - np.var(wb - np.asarray(new_v).flatten()) / np.var(wb))
    return lr_multiplier, info
