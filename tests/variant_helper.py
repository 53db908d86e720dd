"""Run by test_net.py::test_tower_variants_agree in a subprocess (the tower kernel variant is chosen once per
process from CCHESS_TOWER_VARIANT): head-conv outputs of the fused net kernel for a few (blocks, batch) cases."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cchess_zero_amd.net import PolicyValueNet  # noqa: E402

out = sys.argv[1]
res = {}
for blocks, n in ((1, 1), (2, 3), (3, 37), (7, 130)):
    net = PolicyValueNet(blocks, "cuda:0", torch.bfloat16, seed=6, backend="hip")
    gen = torch.Generator().manual_seed(blocks * 1000 + n)
    x = (torch.rand((n, 9, 10, 14), generator=gen) < 0.1).float().cuda()
    res["%d_%d" % (blocks, n)] = net._hip_net_forward(x).cpu()
net = PolicyValueNet(7, "cuda:0", torch.float16, seed=6, backend="hip")   # the default engine's operand type
gen = torch.Generator().manual_seed(7130)
x = (torch.rand((130, 9, 10, 14), generator=gen) < 0.1).float().cuda()
res["f16_7_130"] = net._hip_net_forward(x).cpu()
torch.save(res, out)
