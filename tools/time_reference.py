#!/usr/bin/env python3
"""Times the UNMODIFIED reference (chengstone/cchess-zero main.py, imported from /root/reference through the test
harness) on this container's CPU: MCTS simulations per second of `MCTS_tree.main` with a constant-time fake `forward`
(search-only: move generation, state strings, tree bookkeeping, asyncio) and with the fp32 NumPy restatement of the
2-block / 7-block net as `forward` (end to end).  One process = one core: the reference search is single-threaded
asyncio.  Only runs where /root/reference exists (not on the GPU box); results are recorded in BASELINE.md by hand.

usage: python tools/time_reference.py [playouts=200]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import ref_harness as rh  # noqa: E402
from oracle import net_numpy  # noqa: E402

START = "RNBAKABNR/9/1C5C1/P1P1P1P1P/9/9/p1p1p1p1p/1c5c1/9/rnbakabnr"


def run(forward, playouts, search_threads):
    t = rh.new_mcts(START, forward, search_threads)
    with rh.quiet(), np.errstate(all="ignore"):
        t0 = time.perf_counter()
        t.main(START, "w", 0, playouts)
        dt = time.perf_counter() - t0
    return playouts / dt


def main():
    if "--json" in sys.argv:   # bench.py's cpu_baseline leg: search-only, 50 and 400 playouts, one core
        import json
        const = (np.zeros((1, 2086), np.float32) + 0.01, np.zeros((1, 1), np.float32))
        fake = lambda positions: (np.repeat(const[0], len(positions), 0), np.repeat(const[1], len(positions), 0))
        a = max(run(fake, 50, 16) for _ in range(2))
        b = run(fake, 400, 16)
        print(json.dumps({"search_only_sims_per_s_per_core": b, "search_only_50_playouts_sims_per_s": a, "search_only_400_playouts_sims_per_s": b,
                          "where": "the unmodified reference's MCTS_tree.main imported from /root/reference, constant-time forward, "
                                   "search_threads 16, one core, timed by tools/time_reference.py --json inside this bench run"}))
        return
    playouts = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    const = (np.zeros((1, 2086), np.float32) + 0.01, np.zeros((1, 1), np.float32))

    def fake(positions):
        n = len(positions)
        return np.repeat(const[0], n, 0), np.repeat(const[1], n, 0)
    for st in (1, 16):
        print("reference search-only, constant forward, search_threads=%d, playouts=%d: %.0f sims/s (1 core)" % (st, playouts, run(fake, playouts, st)))
    from cchess_zero_amd.net import PolicyValueModule
    for blocks in (2, 7):
        w = PolicyValueModule(blocks, seed=0).export_tf_layout()

        def fwd(positions, w=w, blocks=blocks):
            return net_numpy.forward(w, np.asarray(positions, np.float32), blocks)
        print("reference end to end, NumPy fp32 %d-block net, search_threads=16, playouts=%d: %.1f sims/s (1 core + BLAS threads)" % (blocks, min(playouts, 64), run(fwd, min(playouts, 64), 16)))


if __name__ == "__main__":
    main()
