#coding:utf-8
"""Drop-in for the reference's policy_value_network.py — same class name and method signatures
(forward / train_step / save / restore / train_restore), MI355X-native underneath.

  reference (TF1 graph + session)                       here
  policy_value_network.py:45-74,151-162  graph          cchess_zero_amd/net.py (PyTorch-ROCm), tower convs by
                                                         the fused MFMA kernel cz_tower_c128_bf16 (csrc/)
  :202-214  forward(positions)->(logits[B,2086], v[B,1]) identical signature; ndarray or list of [9,10,14]
  :77-126,186-199  loss / Nesterov-momentum SGD / clip   train_step(): CE + MSE + 1e-4*sum(w^2)/2, momentum 0.9,
                                                         use_nesterov, clip_by_global_norm(100), NaN check
  :164-184  tf.train.Saver, ./models/best_model.ckpt-N   torch checkpoints under the same directory/prefix
There is no CPU fallback: constructing the network without a HIP device raises.
"""
import glob
import os
import re

import numpy as np
import torch
import torch.nn.functional as F

from cchess_zero_amd.net import PolicyValueModule, PolicyValueNet


class policy_value_network(object):
    def __init__(self, res_block_nums=7, device=None, dtype=torch.bfloat16, save_dir="./models", seed=0):
        if not torch.cuda.is_available():
            raise RuntimeError("policy_value_network needs an MI355X (HIP) device; the cchess_hip path has no CPU fallback")
        self.save_dir = save_dir
        self.is_logging = True
        self.filters_size = 128
        self.prob_size = 2086
        self.c_l2 = 0.0001
        self.momentum = 0.9
        self.global_norm = 100
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self.net = PolicyValueNet(res_block_nums, self.device, dtype, seed=seed)
        self.module = self.net.module
        self.global_step = 0
        self._opt = None
        self.train_restore()

    # ---- inference --------------------------------------------------------------------------
    def forward(self, positions):
        """positions: ndarray [B,9,10,14] or list of [9,10,14] -> (logits [B,2086] f32, value [B,1] f32)."""
        return self.net.forward(positions)

    def forward_device(self, planes):
        """Device-resident variant used by the batched search loop."""
        return self.net.forward_device(planes)

    # ---- training (policy_value_network.py:77-126,186-199) -------------------------------------------
    def _optimizer(self, lr):
        if self._opt is None:
            self._opt = torch.optim.SGD(self.module.parameters(), lr=lr, momentum=self.momentum, nesterov=True)
        for g in self._opt.param_groups:
            g["lr"] = lr
        return self._opt

    def loss(self, positions, probs, winners, training=True):
        x = torch.as_tensor(np.asarray(positions, dtype=np.float32)).to(self.device).permute(0, 3, 1, 2)
        pi = torch.as_tensor(np.asarray(probs, dtype=np.float32)).to(self.device)
        z = torch.as_tensor(np.asarray(winners, dtype=np.float32)).to(self.device).reshape(-1, 1)
        logits, v = self.module(x, training=training)
        policy_loss = -(pi * F.log_softmax(logits, dim=1)).sum(dim=1).mean()   # softmax_cross_entropy_with_logits
        value_loss = F.mse_loss(v, z)                                          # tf.losses.mean_squared_error
        l2 = sum((p * p).sum() for p in self.module.parameters()) * (self.c_l2 / 2.0)  # l2_regularizer over ALL trainables
        accuracy = (logits.argmax(dim=1) == pi.argmax(dim=1)).float().mean()
        return value_loss + policy_loss + l2, accuracy

    def train_step(self, positions, probs, winners, learning_rate):
        """-> (accuracy, loss, global_step), like policy_value_network.py:186-199."""
        opt = self._optimizer(float(learning_rate))
        self.module.train()
        opt.zero_grad(set_to_none=True)
        loss, accuracy = self.loss(positions, probs, winners, training=True)
        loss.backward()
        from cchess_zero_amd.parallel import allreduce_gradients
        allreduce_gradients(self.module)                                       # no-op without a process group
        torch.nn.utils.clip_grad_norm_(self.module.parameters(), self.global_norm)   # tf.clip_by_global_norm
        for p in self.module.parameters():                                     # tf.check_numerics('NaN Found!')
            if p.grad is not None and not torch.isfinite(p.grad).all():
                raise FloatingPointError("NaN Found!")
        opt.step()
        self.module.eval()
        self.net.refresh()
        self.global_step += 1
        return float(accuracy.detach()), float(loss.detach()), self.global_step

    # ---- checkpoints (policy_value_network.py:164-184) ----------------------------------------------------
    def _ckpts(self):
        out = []
        for f in glob.glob(os.path.join(self.save_dir, "best_model.ckpt-*.pt")):
            m = re.search(r"ckpt-(\d+)\.pt$", f)
            if m:
                out.append((int(m.group(1)), f))
        return sorted(out)

    def train_restore(self):
        if not os.path.isdir(self.save_dir):
            os.makedirs(self.save_dir, exist_ok=True)
        c = self._ckpts()
        if c:
            self.restore(c[-1][1])
            print("Successfully loaded:", c[-1][1])
        else:
            print("Could not find old network weights")

    def restore(self, file):
        print("Restoring from {0}".format(file))
        d = torch.load(file, map_location=self.device)
        self.module.load_state_dict(d["model"])
        self.global_step = int(d.get("global_step", 0))
        self.net.refresh()

    def save(self, in_global_step):
        os.makedirs(self.save_dir, exist_ok=True)
        path = os.path.join(self.save_dir, "best_model.ckpt-%d.pt" % int(in_global_step))
        torch.save({"model": self.module.state_dict(), "global_step": int(in_global_step)}, path)
        print("Model saved in file: {}".format(path))
        return path
