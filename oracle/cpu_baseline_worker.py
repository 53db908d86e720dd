#!/usr/bin/env python3
"""One worker process of bench.py's cpu_baseline leg (TEST INFRASTRUCTURE, never on the product path):
`games` trees searched in lock-step by the C oracle (the pinned restatement of the reference's search), leaves
evaluated in one batch by the fp32 torch module on `threads` CPU threads pinned to this worker's own cores.

    python oracle/cpu_baseline_worker.py idx games threads blocks seconds first_core   ->  one JSON line
"""
import json
import os
import sys
import time

idx, games, threads, blocks, seconds, first_core = (int(float(a)) if i != 4 else float(a) for i, a in enumerate(sys.argv[1:7]))
# affinity and thread-count environment BEFORE torch / OpenMP are loaded: an OpenMP runtime that binds its pool from the
# process mask it sees at start-up would otherwise put every worker's threads on the same cores
try:
    avail = sorted(os.sched_getaffinity(0))
    mine = avail[first_core:first_core + threads] or avail[:threads]
    os.sched_setaffinity(0, mine)
except Exception:
    mine = []
for k in ("GOMP_CPU_AFFINITY", "KMP_AFFINITY", "OMP_PLACES"):
    os.environ.pop(k, None)
os.environ.update(OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), OMP_PROC_BIND="false",
                  HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch as T  # noqa: E402

T.set_num_threads(threads)
from oracle import oracle as O  # noqa: E402
from cchess_zero_amd.net import PolicyValueModule  # noqa: E402

START = O.fen_to_board(O.START_FEN)
rng = np.random.default_rng(idx)
boards = np.tile(START, (games, 1))
side = np.zeros(games, np.uint8)
for g in range(games):  # short random playouts with the oracle: the same position family as the GPU bench
    b, s = boards[g].copy(), 0
    for _ in range(int(rng.integers(0, 60))):
        mv = O.legal_moves(b, s)
        if len(mv) == 0:
            break
        nb, cap, term = O.apply_move(b, int(mv[rng.integers(len(mv))]))
        if term:
            break
        b, s = nb, s ^ 1
    boards[g], side[g] = b, s
m = PolicyValueModule(blocks, seed=0).eval()
srch = O.Search(games, 40000)
srch.reset(boards, side, None)
t0 = time.perf_counter()
sims, step, t_net = 0, 0, 0.0
with T.no_grad():
    while True:
        planes, need = srch.select(0 if step == 0 else 1)
        t1 = time.perf_counter()
        lg, v = m(T.from_numpy(planes).permute(0, 3, 1, 2))
        t_net += time.perf_counter() - t1
        srch.expand_backup(lg.numpy(), v.numpy())
        if step > 0:
            sims += games
        step += 1
        if time.perf_counter() - t0 > seconds and step > 2:
            break
dt = time.perf_counter() - t0
print(json.dumps({"sims": sims, "seconds": dt, "steps": step - 1, "net_seconds": t_net, "cores": mine[:4] + (["..."] if len(mine) > 4 else [])}))
