#!/usr/bin/env python3
"""bench.py — MCTS simulations/sec of the MI355X hot path (select -> batched net -> expand/backup).

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched through
torch.distributed.run, one rank per GPU.  One "step" = one lock-step simulation of EVERY game tree on
the rank (G simulations); after every `--playout` simulations the trees advance one ply
(root-visit argmax -> cz_search_advance, the update_tree of the reference) so long runs stay in the
self-play regime.  Rank 0 prints ONE JSON line.  Metric/config follow BASELINE.json:
"MCTS simulations/sec (whole node), playout=1600, 7-block net", 8192 games per GPU (configs[2]).

Inputs are synthetic: seeded random-playout positions generated on the GPU with the rules kernels,
Glorot-uniform weights (seed 0).  Nothing here reads /root/reference.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3}  # dense peaks, same guide

START = np.array([3, 5, 4, 2, 1, 2, 4, 5, 3] + [0] * 9 + [0, 7, 0, 0, 0, 0, 0, 7, 0] + [6, 0, 6, 0, 6, 0, 6, 0, 6] + [0] * 18 +
                 [13, 0, 13, 0, 13, 0, 13, 0, 13] + [0, 14, 0, 0, 0, 0, 0, 14, 0] + [0] * 9 + [10, 12, 11, 9, 8, 9, 11, 12, 10], np.uint8)


def synth_positions(rules, G, seed, max_ply=80):
    """Seeded uniform-random playouts from the start position, ply ~ U[0, max_ply] per game, all on the GPU
    (K1 movegen -> random pick -> K2 apply).  Games whose king would be captured stop early."""
    dev = rules.dev
    gen = torch.Generator(device=dev).manual_seed(seed)
    boards = torch.from_numpy(np.tile(START, (G, 1))).to(dev)
    side = torch.zeros(G, dtype=torch.uint8, device=dev)
    rr = torch.zeros(G, dtype=torch.int32, device=dev)
    target = torch.randint(0, max_ply + 1, (G,), generator=gen, device=dev)
    alive = torch.ones(G, dtype=torch.bool, device=dev)
    for ply in range(max_ply):
        moves, count, _ = rules.movegen(boards, side, want_mask=False)
        cnt = count.to(torch.int64) & 0xFFFF
        go = alive & (target > ply) & (cnt > 0)
        r = (torch.rand(G, generator=gen, device=dev) * cnt.clamp(min=1)).to(torch.int64).clamp(max=127)
        pick = moves.gather(1, r.unsqueeze(1)).squeeze(1)
        # do not play a king capture: keep both kings on the board for the search roots
        nb, ns = boards.clone(), side.clone()
        lab = torch.where(go, pick, torch.full_like(pick, -1))
        cap, term = rules.apply_move(nb, ns, lab)
        ok = go & (term == 0)
        boards = torch.where(ok.unsqueeze(1), nb, boards)
        side = torch.where(ok, ns, side)
        rr = torch.where(ok, torch.where(cap != 0, torch.zeros_like(rr), rr + 1), rr)
        alive = alive & (ok | ~go)
    return boards.contiguous(), side.contiguous(), rr.contiguous()


def cpu_baseline(blocks, seconds_target=15.0):
    """CPU port timed on this host: C oracle search (oracle/) + NumPy fp32 net restatement, a bounded
    sample of the same workload (same position generator family, playout-style lock-step).  Test
    infrastructure used as the *baseline being measured*, never as the product path."""
    from oracle import oracle as O
    from oracle import net_numpy
    from cchess_zero_amd.net import PolicyValueModule
    try:
        from threadpoolctl import threadpool_limits
    except Exception:  # pragma: no cover
        threadpool_limits = None
    cores = os.cpu_count() or 1
    G = 128   # enough rows per net call to keep the BLAS threads of a many-core host busy
    rng = np.random.default_rng(0)
    boards = np.tile(START, (G, 1))
    side = np.zeros(G, np.uint8)
    for g in range(G):  # short random playouts with the oracle
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
    w = PolicyValueModule(blocks, seed=0).export_tf_layout()
    s = O.Search(G, 20000)
    s.reset(boards, side, None)
    ctxm = threadpool_limits(limits=cores) if threadpool_limits else None
    t0 = time.perf_counter()
    sims = 0
    step = 0
    while True:
        planes, need = s.select(0 if step == 0 else 1)
        logits, v = net_numpy.forward(w, planes, blocks)
        s.expand_backup(logits, v)
        if step > 0:
            sims += G
        step += 1
        if time.perf_counter() - t0 > seconds_target and step > 2:
            break
    dt = time.perf_counter() - t0
    if ctxm is not None:
        ctxm.__exit__(None, None, None)
    return {"value": sims / dt, "unit": "sims/s", "cores": cores, "kind": "port",
            "sample": "%d games x %d lock-step simulations, C oracle search + NumPy fp32 %d-block net, %.1f s" % (G, step - 1, blocks, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1600)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--games", type=int, default=8192, help="game trees per GPU")
    ap.add_argument("--playout", type=int, default=1600)
    ap.add_argument("--blocks", type=int, default=7)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--backend", default="auto", choices=["auto", "hip", "torch"], help="conv backend of the net")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"], help="nccl = RCCL (default); gloo only for single-GPU smoke tests of the N>1 path")
    ap.add_argument("--all-on-device0", action="store_true", help="testing only: every rank uses cuda:0")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--search-threads", type=int, default=1, help="simulations in flight per tree and step (the reference's search_threads; virtual loss 3); batch = games * search_threads")
    ap.add_argument("--graph", action="store_true", help="replay the four launches of a step as one captured HIP graph (measured: no gain, the host already runs ahead of the GPU)")
    ap.add_argument("--compact", action="store_true", help="compact evaluation batches: no net row for terminal / drawn leaves (no gain at 8192 trees: the trunk runs in rounds of 1024 rows)")
    ap.add_argument("--full-policy-fc", action="store_true", help="compute all 2086 logits per leaf (k_policy_fc) instead of folding the policy FC into the expansion")
    ap.add_argument("--nodes-per-tree", type=int, default=0, help="node pool capacity per tree (default (playout + 2) * 80: no tree can run out)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.all_on_device0:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    else:
        torch.cuda.set_device(0)
        local_rank = 0
    assert args.gpus == world, "--gpus %d but WORLD_SIZE=%d" % (args.gpus, world)
    dev = torch.device("cuda", local_rank)

    from cchess_zero_amd.engine import Context, SearchEngine
    from cchess_zero_amd.net import PolicyValueNet, flops_per_position
    from cchess_zero_amd.rules import Rules

    G, playout = args.games, args.playout
    tdt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.dtype]
    # node pool per tree: a ply adds ~40 nodes per simulation on top of the subtree kept from the previous ply; 160 per
    # simulation has held over 6+ plies of this workload (80 overflowed 3 000 of 8192 trees by the fourth ply).  Same-box
    # A/B: the capacity does not change the step time (2.300 ms at 128 160 and at 256 320 nodes per tree).  Simulations
    # are counted, so an overflowing (parked) tree could not inflate the result anyway.
    cap = args.nodes_per_tree or (playout + 2) * 160
    ctx = Context(G, cap, local_rank)
    rules = Rules(ctx)
    # planes are written by k_select directly in the fused net kernel's input format (bf16, 16 channels)
    fused = (args.backend in ("auto", "hip")) and args.dtype in ("bf16", "fp16")
    K = max(1, args.search_threads)
    eng = SearchEngine(G, cap, local_rank, plane_dtype=tdt if fused else torch.float32, channels=16 if fused else 14, ctx=ctx, width=K)
    net = PolicyValueNet(args.blocks, dev, tdt, seed=0, backend=args.backend, ctx=ctx)
    fused_fc = net.fused_search and not args.full_policy_fc and K == 1
    compact = fused_fc and args.compact   # leaves that need no net evaluation are not in the net's batch
    boards, side, rr = synth_positions(rules, G, seed=1000 + rank)
    eng.reset(boards, side, rr)

    ev_net = []  # (start, end) events around the net forward of every timed step
    conv_ev = []  # (start, end) events around single launches of the dominant kernel

    step_no = [0]

    def one_step(mode, timed):
        # HIP events around every conv launch of every 8th timed step (same stream as the launches)
        net.conv_events = conv_ev if (timed and step_no[0] % 8 == 0) else None
        step_no[0] += 1
        n_rows = None
        if compact:
            planes, n_rows = eng.select_compact(mode)
        else:
            planes, _ = eng.select(mode)
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if fused_fc:   # trunk + value head; the policy FC runs inside the expansion kernel for the legal moves only
            z, value = net.search_eval(planes, n_rows)
        else:
            logits, value = net.forward_device(planes)
        if timed:
            e1.record()
            ev_net.append((e0, e1))
        if fused_fc:
            eng.expand_backup_fc(z, value, net.pfc_w_rows, net.pfc_b_f32, compact=compact)
        else:
            eng.expand_backup(logits, value)

    gather_ok = None   # N > 1: result of the (untimed) record all-gather
    banked = [0]   # simulations completed in plies that were closed inside the timed region (k > 1 accounting)

    def advance_ply():
        banked[0] += int(eng.status()[2].sum().item())   # one sync per ply (every `playout` steps)
        st = eng.root_stats()
        n = st["N"].clone()
        cnt = (st["count"].to(torch.int64) & 0xFFFF).unsqueeze(1)
        n[torch.arange(128, device=dev).unsqueeze(0) >= cnt] = -1
        best = n.argmax(dim=1, keepdim=True)
        played = st["label"].gather(1, best).squeeze(1)
        eng.advance(played)
        one_step(0, False)  # expand roots that were never visited

    sims_in_ply = 0

    graph = [None]   # the steady-state step (4 launches, static arguments) captured as one HIP graph

    def run(nsteps, timed):
        nonlocal sims_in_ply
        for _ in range(nsteps):
            if sims_in_ply >= playout:
                advance_ply()
                sims_in_ply = 0
            # every 8th timed step runs eagerly so that HIP events can bracket the trunk launch on its stream
            if graph[0] is not None and not (timed and step_no[0] % 8 == 0):
                step_no[0] += 1
                graph[0].replay()
            else:
                one_step(1, timed)
            sims_in_ply += K

    one_step(0, False)          # MCTS_tree.main root expansion (not a simulation)
    run(args.warmup, False)
    torch.cuda.synchronize()
    if args.graph and fused_fc and not compact:
        try:   # capture AFTER the warm-up (kernel attributes set, allocator warm); a failure falls back to eager launches
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                one_step(1, False)
            step_no[0] -= 1                      # the capture itself executed nothing
            torch.cuda.synchronize()
            graph[0] = gr
        except Exception as e:
            print("bench: HIP graph capture failed (%r), running eagerly" % (e,), file=sys.stderr)
            graph[0] = None
            torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    sims0 = int(eng.status()[2].sum().item())
    rows0, csteps0 = eng.eval_totals() if compact else (0, 0)
    banked[0] = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps, True)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist_on:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        # outside the timed region: the record exchange of the self-play loop (one all-gather of packed
        # (s, pi, z) records over RCCL) on a token batch, so the N>1 run exercises the collective path too
        from cchess_zero_amd import parallel, selfplay
        tok = np.zeros((4 + rank, selfplay.REC_BYTES), np.uint8)
        tok[:, 0] = rank + 1
        try:
            allrec = parallel.gather_records(tok, device=dev if args.dist_backend == "nccl" else "cpu")
            gather_ok = bool(allrec.shape[0] == sum(4 + r for r in range(world)) and int(allrec[-1, 0]) == world)
        except Exception as e:   # the throughput number above must survive a failure of this (untimed) exchange
            gather_ok = "failed: %r" % (e,)

    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process; the committed
    # rocprofv3 --pmc measurement of the same launch shape is attached when the configuration matches.
    tower_kernel = "k_tower_c128" if os.environ.get("CCHESS_TOWER_VARIANT", "") == "4w" else "k_tower8_c128"
    traffic, traffic_src = None, None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        want = {"B": G, "res_block_nums": args.blocks, "dtype": args.dtype}
        if ({k: tj["config"].get(k) for k in want} == want and bool(tj["config"].get("compact", False)) == bool(compact)
                and args.backend in ("auto", "hip") and tj["kernel"] == tower_kernel):
            traffic = tj["traffic_bytes_per_launch"]
            traffic_src = "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE, separate passes, FETCH x2 gfx950 correction; algorithmic bytes %d)" % tj["algorithmic_bytes_per_launch"]
    except Exception:
        pass
    st, nodes, sims, depth = eng.status()
    bad = int((st & ~8).ne(0).sum().item())
    st_bits = {name: int(((st & bit) != 0).sum().item()) for name, bit in
               (("pool_exhausted", 1), ("no_moves", 2), ("move_overflow", 4), ("bad_advance", 8))}
    net_ms = float(np.mean([a.elapsed_time(b) for a, b in ev_net])) if ev_net else float("nan")
    # simulations are COUNTED (completed backups, per-tree device counters), not assumed: a parked tree (node pool
    # exhausted) or an abandoned descent (k > 1) contributes nothing.  With k = 1 and no parked tree this is G * steps.
    mine = torch.tensor([banked[0] + int(sims.sum().item()) - sims0], dtype=torch.float64, device=dev if (dist_on and args.dist_backend == "nccl") else "cpu")
    if dist_on:
        dist.all_reduce(mine)
    total_sims = float(mine.item())
    # rows the net evaluated per launch: all G without compaction, else the measured mean over the timed region
    rows_per_launch = float(G)
    if compact:
        rows1, csteps1 = eng.eval_totals()
        rows_per_launch = (rows1 - rows0) / max(1, csteps1 - csteps0)
    flops = flops_per_position(args.blocks) * rows_per_launch
    peak = MFMA_PEAK_TFLOPS[args.dtype]
    if conv_ev:
        # dominant kernel: k_conv3x3_c128 (one launch = one fused tower layer over the whole batch)
        conv_ms = float(np.mean([a.elapsed_time(b) for a, b in conv_ev]))
        nl = 2 * args.blocks if net.backend == "hip" else 1   # fused tower: one launch = all 2*blocks conv layers
        conv_flops = 2.0 * rows_per_launch * 90 * 1152 * 128 * nl
        kname = (tower_kernel + " (first conv + whole residual tower + head 1x1 convs in one launch: %d conv3x3+BN(+residual)+ReLU layers counted, LDS-resident activations, %s MFMA, fp32 acc)" % (nl, args.dtype)
                 if net.backend == "hip" else "k_conv3x3_c128 (fused conv3x3+BN+residual+ReLU, bf16 MFMA, fp32 acc)")
        roof = {"bound": "mfma", "kernel": kname,
                "achieved": conv_flops / (conv_ms * 1e-3) / 1e12, "peak": peak, "unit": "TFLOP/s",
                "frac": conv_flops / (conv_ms * 1e-3) / 1e12 / peak, "traffic": traffic, "traffic_source": traffic_src,
                "us_per_launch": conv_ms * 1e3, "launches_timed": len(conv_ev), "flops_per_launch": conv_flops,
                "net_forward_ms_per_step": net_ms, "net_forward_tflops": flops / (net_ms * 1e-3) / 1e12}
    else:
        achieved = flops / (net_ms * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": "net forward via torch/MIOpen (conv tower + heads), all launches of one step",
                "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": None,
                "ms_per_launch_group": net_ms, "flops_per_step": flops}
    if (G, playout, args.blocks) == (8192, 1600, 7):
        cfg_name = "BASELINE.json configs[2]" + ("; per-GPU share of configs[3]" if world > 1 else "")
    elif (G, playout, args.blocks, args.dtype) == (8192, 1600, 19, "fp16"):
        cfg_name = "per-GPU share of BASELINE.json configs[4]"
    elif (G, playout, args.blocks) == (4096, 400, 7):
        cfg_name = "BASELINE.json configs[1]"
    else:
        cfg_name = "custom configuration"
    out = {
        "metric": "MCTS simulations/sec (whole node), playout=%d, %d-block net" % (playout, args.blocks),
        "value": total_sims / dt, "unit": "sims/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "%d parallel games per GPU, playout=%d, %d-block net %s (%s)" % (G, playout, args.blocks, args.dtype, cfg_name),
                   "games_per_gpu": G, "playout": playout, "conv_backend": net.backend, "policy_fc": "in-expansion, legal moves only" if fused_fc else "full 2086 logits",
                   "net_rows_per_step": rows_per_launch, "compact_batches": bool(compact), "hip_graph": graph[0] is not None, "record_gather": gather_ok, "res_block_nums": args.blocks, "search_threads": K,
                   "positions": "seeded random playouts from the start position, ply~U[0,80]",
                   "simulations_counted": total_sims, "simulations_nominal": float(G) * args.steps * world * K,
                   "mean_leaf_depth": float(depth.float().mean().item()), "mean_nodes_per_tree": float(nodes.float().mean().item()),
                   "trees_with_error_status": bad, "status_bits": st_bits},
        "roofline": roof,
    }
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.blocks, args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
