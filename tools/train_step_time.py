#!/usr/bin/env python3
"""Time Trainer.train_step (7 blocks, batch 512, fp32 torch module on cuda:0) with MIOpen restricted to deterministic
convolution kernels (the product setting, train.deterministic_convolutions) and unrestricted.  argv[1] = det | nondet."""
import contextlib
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import cchess_zero_amd.train as T
from cchess_zero_amd.net import PolicyValueModule

if len(sys.argv) > 1 and sys.argv[1] == "nondet":
    T.deterministic_convolutions = contextlib.nullcontext
m = PolicyValueModule(7, seed=1).to("cuda:0")
tr = T.Trainer(m)
rng = np.random.default_rng(0)
x = (rng.random((512, 9, 10, 14)) < 0.07).astype(np.float32)
pi = rng.random((512, 2086)).astype(np.float32)
pi /= pi.sum(1, keepdims=True)
z = rng.choice([-1.0, 1.0], (512, 1)).astype(np.float32)
for _ in range(3):
    tr.train_step(x, pi, z, 0.001)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    tr.train_step(x, pi, z, 0.001)
torch.cuda.synchronize()
print("%s: %.2f ms per train_step (7 blocks, batch 512)" % (sys.argv[1] if len(sys.argv) > 1 else "det", (time.perf_counter() - t0) / 20 * 1e3))
