#coding:utf-8
"""Drop-in for the reference's policy_value_network.py — same class name and method signatures
(forward / train_step / save / restore / train_restore), MI355X-native underneath.

  reference (TF1 graph + session)                       here
  policy_value_network.py:45-74,151-162  graph          cchess_zero_amd/net.py (PyTorch-ROCm), tower convs by
                                                         the fused MFMA kernel cz_net_trunk_f16 (csrc/)
  :202-214  forward(positions)->(logits[B,2086], v[B,1]) identical signature; ndarray or list of [9,10,14]
  :77-126,186-199  loss / Nesterov-momentum SGD / clip   cchess_zero_amd/train.py Trainer: CE + MSE + 1e-4*sum(w^2)/2,
                                                         momentum 0.9, use_nesterov, clip_by_global_norm(100), NaN check
  :164-184  tf.train.Saver, ./models/best_model.ckpt-N   torch checkpoints under the same directory/prefix (weights,
                                                         momentum slots, global step); restore() also takes an .npz of
                                                         the reference's TF1 variables (net.from_tf_variables)
There is no CPU fallback: constructing the network without a HIP device raises.
"""
import glob
import os
import re

import numpy as np
import torch

from cchess_zero_amd import tf_checkpoint
from cchess_zero_amd.net import (PolicyValueModule, PolicyValueNet, from_tf_variables, momentum_slots_from_tf_variables,
                                 to_tf_variables)
from cchess_zero_amd.train import Trainer


class policy_value_network(object):
    PRECISIONS = {"strict": (torch.float16, "strict"), "mx6": (torch.float16, "mx"), "fp16x2": (torch.float16, True),
                  "bf16x2": (torch.bfloat16, True), "fp16": (torch.float16, False), "bf16": (torch.bfloat16, False),
                  "fp32": (torch.float32, False)}

    def __init__(self, res_block_nums=7, device=None, dtype=None, save_dir="./models", seed=0, precision=None):
        """precision (or the environment's CCHESS_NET_PRECISION; default "strict"): which engine evaluates the net.
          "strict"   the drop-in default: forward() within 1e-3 ABSOLUTE of the reference's fp32 sess.run
                     (policy_value_network.py:202-214), MEASURED, not assumed: it starts on "mx6" (on "bf16x2" for bf16 nets)
                     and after every weight change (construction, restore(), each train_step) the engine and
                     the fp32 module evaluate 64 distinct positions; above 5e-4 in a logit or the value the net falls over
                     mx6 -> fp16x2 -> fp32 and says so (net.strict_report / strict_report() hold the last measurement);
          "mx6"      fp16 hi halves on fp16 MFMAs + both cross terms of the hi + lo split on ONE block-scaled fp6 MFMA
                     (k_trunk_mx_c128, 1.5 MFMA-equivalents per product): 5e-4 / 2e-4 at 7 blocks, 1.0e-3 / 1.1e-3 at 19;
                     1.9 M simulations/s;
          "fp16x2"   every weight and stored activation as fp16 hi + lo halves, three MFMAs per product
                     (k_trunk_split_c128): 7e-5 / 2e-4 also at 19 blocks; 1.3 M simulations/s;
          "fp16"     one fp16 per operand (k_tower8_c128): 3.6-3.9 M simulations/s, 1.2e-4 on TF-default weights, 1.3e-2
                     absolute (1.1e-3 of the largest logit) on peaked weights;
          "bf16"     the same rate + 3 %, 8x the error;   "bf16x2": fp16x2 with bf16 halves (2e-4 / 7e-4);
          "fp32"     torch/MIOpen fp32 (1e-7 .. 3e-5), not a kernel of this library.
        dtype (older callers): a torch dtype selects the one-value-per-operand engine of that type."""
        if dtype is not None:
            split = False
        else:
            name = precision or os.environ.get("CCHESS_NET_PRECISION", "strict")
            if name not in self.PRECISIONS:
                raise ValueError("precision must be one of %s" % sorted(self.PRECISIONS))
            dtype, split = self.PRECISIONS[name]
        if not torch.cuda.is_available():
            raise RuntimeError("policy_value_network needs an MI355X (HIP) device; the cchess_hip path has no CPU fallback")
        self.save_dir = save_dir
        self.is_logging = True
        self.filters_size = 128
        self.prob_size = 2086
        self.c_l2 = 0.0001
        self.momentum = 0.9
        self.global_norm = 100
        self.max_to_keep = 5
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self.net = PolicyValueNet(res_block_nums, self.device, dtype, seed=seed, split=split)
        self.module = self.net.module
        self.trainer = Trainer(self.module, self.c_l2, self.momentum, self.global_norm)
        self.train_restore()

    @property
    def global_step(self):
        return self.trainer.global_step

    @global_step.setter
    def global_step(self, v):
        self.trainer.global_step = int(v)

    def refresh(self):
        self.net.refresh()

    def strict_report(self):
        """precision "strict": the last self-measurement of the engine against fp32 on the live weights ({"engine", "dlogit",
        "dvalue", "max_abs_logit", "tol", "positions", "fell_over_from"}; measured now if a weight change is pending)."""
        if self.net.strict_auto:
            self.net._ensure_checked()
        return self.net.strict_report

    # ---- inference --------------------------------------------------------------------------
    def forward(self, positions):
        """positions: ndarray [B,9,10,14] or list of [9,10,14] -> (logits [B,2086] f32, value [B,1] f32)."""
        return self.net.forward(positions)

    def forward_device(self, planes):
        """Device-resident variant used by the batched search loop."""
        return self.net.forward_device(planes)

    # ---- training (policy_value_network.py:77-126,186-199) -------------------------------------------
    def loss(self, positions, probs, winners, training=True):
        return self.trainer.loss(positions, probs, winners, training)

    def train_step(self, positions, probs, winners, learning_rate):
        """-> (accuracy, loss, global_step), like policy_value_network.py:186-199."""
        out = self.trainer.train_step(positions, probs, winners, learning_rate)
        self.net.refresh()   # re-fold BN and re-pack the MFMA operands for the new weights
        return out

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
        tf_ckpt = tf_checkpoint.latest_checkpoint(self.save_dir)   # tf.train.get_checkpoint_state(save_dir), :165-168
        # both kinds may sit in one directory (a model directory of the reference that this code has trained on since, or the
        # other way round after export_tf_checkpoint): the NEWEST one — the larger global step — is the model
        tf_step = -1
        if tf_ckpt:
            m = re.search(r"ckpt-(\d+)$", tf_ckpt)
            tf_step = int(m.group(1)) if m else 0
        if c and c[-1][0] >= tf_step:
            self.restore(c[-1][1])
            print("Successfully loaded:", c[-1][1])
        elif tf_ckpt:   # a model directory written by the reference itself (tf.train.Saver: checkpoint + .index + .data)
            self.restore(tf_ckpt)
            print("Successfully loaded:", tf_ckpt)
        else:
            print("Could not find old network weights")

    def restore(self, file):
        """A checkpoint written by save(); or the reference's OWN checkpoint — `best_model.ckpt-N` as tf.train.Saver wrote
        it (V2 tensor bundle: .index + .data-00000-of-00001, read by cchess_zero_amd/tf_checkpoint.py without TensorFlow:
        weights by their TF1 variable names `conv2d/kernel`, `BatchNorm_3/moving_variance`, `fully_connected/weights`, ...,
        the Momentum slots and global_step; HWIO kernels and [in,out] FC weights are converted by load_tf_layout); or an
        .npz holding the same variables by name."""
        print("Restoring from {0}".format(file))
        variables = None
        if str(file).endswith(".npz"):
            variables = np.load(file)
        elif tf_checkpoint.is_tf_checkpoint(file):
            variables = tf_checkpoint.read_checkpoint(file)
        if variables is not None:
            d, blocks, gs = from_tf_variables(variables, self.module.res_block_nums)
            self.module.load_tf_layout(d)
            slots = momentum_slots_from_tf_variables(variables, blocks)
            if slots:
                self.trainer.load_tf_momentum(slots)
            if gs is not None:
                self.global_step = gs
        else:
            self.trainer.load_state_dict(torch.load(file, map_location=self.device))
        self.net.refresh()
        self.net.range_check()   # restored weights: finite outputs, activations inside the fp16 range (else it says so)

    def save(self, in_global_step):
        os.makedirs(self.save_dir, exist_ok=True)
        path = os.path.join(self.save_dir, "best_model.ckpt-%d.pt" % int(in_global_step))
        d = self.trainer.state_dict()
        d["global_step"] = int(in_global_step)
        torch.save(d, path)
        print("Model saved in file: {}".format(path))
        # tf.train.Saver() keeps the five most recent checkpoints (max_to_keep=5, policy_value_network.py:148); with one
        # save per policy update (main.py:1188) anything else fills the disk
        self._saved = [f for f in getattr(self, "_saved", []) if f != path] + [path]
        while len(self._saved) > self.max_to_keep:
            old = self._saved.pop(0)
            try:
                os.remove(old)
            except OSError:
                pass
        return path

    def export_tf_checkpoint(self, in_global_step=None, save_dir=None):
        """The way back: weights, Momentum slots and global_step written as the reference's own tf.train.Saver would
        (`<save_dir>/best_model.ckpt-N.index` + `.data-00000-of-00001` + the `checkpoint` state file,
        policy_value_network.py:176-184), under the reference graph's TF1 variable names — the reference's restore() /
        train_restore() load it into its TF graph.  Returns the checkpoint prefix."""
        step = int(self.global_step if in_global_step is None else in_global_step)
        d = to_tf_variables(self.module, None)
        d["global_step"] = np.asarray(step, np.int32)
        tf_names = {k: v for k, v in __import__("cchess_zero_amd.net", fromlist=["tf_variable_names"]).tf_variable_names(self.module.res_block_nums).items()}
        slots = self.trainer.tf_momentum_slots()
        for tf_name, ours in tf_names.items():
            if ours in slots:
                d[tf_name + "/Momentum"] = slots[ours]
        out_dir = save_dir or self.save_dir
        os.makedirs(out_dir, exist_ok=True)
        name = "best_model.ckpt-%d" % step
        tf_checkpoint.write_checkpoint(os.path.join(out_dir, name), d)
        tf_checkpoint.write_checkpoint_state(out_dir, name)
        return os.path.join(out_dir, name)

    def export_tf_variables(self, file=None):
        """The weights under the reference graph's TF1 variable names (optionally written as .npz)."""
        d = to_tf_variables(self.module, self.global_step)
        if file is not None:
            np.savez(file, **d)
        return d
