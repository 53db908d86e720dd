import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from cchess_zero_amd.engine import SearchEngine
from cchess_zero_amd.net import PolicyValueNet
from cchess_zero_amd.rules import START_BOARD
net = PolicyValueNet(2, "cuda:0", torch.float16, seed=5, split="strict")
G, K, playouts = 3, 16, 96
boards, side = np.tile(START_BOARD, (G, 1)), np.zeros(G, np.uint8)
e = SearchEngine(G, 20000, plane_dtype=torch.float32, channels=14, width=K)
e.reset(boards, side, None)
e.step(net.forward_device, mode=0)
print("after mode 0:", [x.cpu().numpy() for x in e.status()])
for i in range(12):
    e.step(net.forward_device, mode=1)
    st, nodes, sims, depth = [x.cpu().numpy() for x in e.status()]
    print(i, "status", st, "nodes", nodes, "sims", sims, "depth", depth, "needs", int(e.need.sum()), "N sum", e.root_stats_host()["N"].sum(axis=1))
lg, v = net.forward_device(e.planes)
print("logits range", float(lg.min()), float(lg.max()), "value", v.flatten()[:5])
