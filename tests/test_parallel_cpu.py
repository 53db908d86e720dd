"""N>1 path on CPU: world_size-2 gloo processes shard games by rank and all-gather packed (s, pi, z)
records; record packing round-trips and expands to the reference's dense training tuples."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fake_records(rank, n, seed=0):
    """Records produced by playing random games with the C oracle (stand-in for the GPU self-play)."""
    sys.path.insert(0, ROOT)
    from oracle import oracle as O
    from cchess_zero_amd import selfplay as SP
    rng = np.random.default_rng(seed + 17 * rank)
    B, S, L, P, C, Z = [], [], [], [], [], []
    b, s = O.fen_to_board(O.START_FEN), 0
    for i in range(n):
        mv = O.legal_moves(b, s)
        k = len(mv)
        l2 = np.full(128, 0xFFFF, np.uint16); l2[:k] = mv
        p2 = np.zeros(128, np.int64); p2[:k] = rng.integers(0, 400, k)   # root visit counts (some zero)
        p2[int(rng.integers(k))] += 1 + int(rng.integers(1600))
        B.append(b.copy()); S.append(s); L.append(l2); P.append(p2); C.append(k); Z.append(int(rng.integers(-1, 2)))
        nb, cap, term = O.apply_move(b, int(mv[rng.integers(k)]))
        if term:
            b, s = O.fen_to_board(O.START_FEN), 0
        else:
            b, s = nb, s ^ 1
    return SP.pack_records(np.stack(B), np.asarray(S, np.uint8), np.stack(L), np.stack(P), np.asarray(C, np.uint8), np.asarray(Z, np.int8))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cchess_zero_amd import parallel as PL
    games = PL.shard_games(11, rank, world)
    rec = _fake_records(rank, 5 + 3 * rank)         # ragged shards: 5 and 8 records
    allrec = PL.gather_records(rec)
    empty = PL.gather_records(np.zeros((0, rec.shape[1]), np.uint8) if rank == 0 else rec)  # one empty shard
    # weight broadcast + gradient all-reduce
    lin = torch.nn.Linear(4, 3)
    torch.manual_seed(rank)
    with torch.no_grad():
        lin.weight.fill_(float(rank + 1))
    PL.broadcast_weights(lin, src=0)
    lin.weight.grad = torch.full_like(lin.weight, float(rank))
    lin.bias.grad = torch.full_like(lin.bias, float(2 * rank))
    PL.allreduce_gradients(lin)
    q.put((rank, games.tolist(), rec.tobytes(), allrec.tobytes(), allrec.shape, empty.shape,
           float(lin.weight[0, 0]), float(lin.weight.grad[0, 0]), float(lin.bias.grad[0])))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_shard_and_gather():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, g0, rec0, all0, shape0, e0, w0, gw0, gb0), (r1, g1, rec1, all1, shape1, e1, w1, gw1, gb1) = res
    assert g0 == [0, 2, 4, 6, 8, 10] and g1 == [1, 3, 5, 7, 9]          # games sharded g % world
    assert all0 == all1 and shape0 == shape1 == (13, shape0[1])          # same gathered buffer everywhere
    assert all0 == rec0 + rec1                                           # rank order, padding stripped
    assert e0 == e1 == (8, shape0[1])                                    # an empty shard is fine
    assert w0 == w1 == 1.0                                               # broadcast from rank 0
    assert gw0 == gw1 == 0.5 and gb0 == gb1 == 1.0                       # averaged gradients


def test_record_roundtrip_and_dense_expansion():
    sys.path.insert(0, ROOT)
    from oracle import oracle as O
    from cchess_zero_amd import selfplay as SP
    from cchess_zero_amd import _lib
    rec = _fake_records(0, 40, seed=3)
    u = SP.unpack_records(rec)
    assert rec.shape[1] == SP.REC_BYTES
    planes, pi, z = SP.to_dense(rec)
    unflip = _lib.tables()["unflip"]
    for i in range(len(z)):
        k = int(u["counts"][i])
        assert np.array_equal(u["labels"][i, :k], O.legal_moves(u["boards"][i], int(u["side"][i])))
        # planes == generate_inputs of the recorded position (oracle = reference restatement)
        assert np.array_equal(planes[i], O.encode_planes(u["boards"][i], int(u["side"][i])))
        # pi lives on canonical labels: rank-flipped for black (main.py:1507-1512)
        lab = u["labels"][i, :k].astype(np.int64)
        if u["side"][i]:
            lab = unflip[lab]
        # pi = the reference's softmax(1/T * log(visits)) (main.py:1341), bit for bit from the stored visit counts
        v = tuple(int(x) for x in u["visits"][i, :k])
        with np.errstate(divide="ignore"):
            x = 1.0 / 1 * np.log(v)
        probs = np.exp(x - np.max(x))
        probs /= np.sum(probs)
        assert np.array_equal(pi[i, lab], probs) and abs(float(pi[i].sum()) - 1.0) < 1e-12
        assert np.count_nonzero(pi[i]) == np.count_nonzero(v)
        assert z[i] in (-1.0, 0.0, 1.0)
    # the vectorised expansion agrees to the last bits
    _, pi_fast, _ = SP.to_dense(rec, exact=False)
    assert np.allclose(pi_fast, pi, rtol=1e-14, atol=0)


def _plumbing_worker(rank, world, port, q):
    """bench.py's multi-rank plumbing (cchess_zero_amd/parallel.py), 8 ranks over gloo on the CPU."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cchess_zero_amd import parallel as PL
    from cchess_zero_amd._lib import REC_BYTES
    dev = torch.device("cpu")
    cpus = PL.pin_rank_to_cpus(rank, world)
    slowest = PL.max_over_ranks(1.0 + 0.25 * rank, dev)
    tot = PL.sum_over_ranks([10 * (rank + 1), 2.5], dev)
    pr = PL.per_rank(100.0 + rank, dev)
    ok = PL.gather_selfcheck(dev)
    # fixed-capacity exchange, 6 rounds of ragged shards (0 .. 40 records) through a capacity of 16: every record must
    # arrive exactly once, in per-rank order, whatever was carried over; flush() drains what the capacity held back
    rng = np.random.default_rng(1234)                     # the same shard-size table on every rank
    sizes = rng.integers(0, 41, size=(6, world))
    sizes[2, :] = 0                                       # a round in which nobody has anything
    sizes[3, 1 % world] = 40
    ex = PL.RecordExchange(16, dev)
    sent, got = [], [[] for _ in range(world)]
    serial = 0
    def take(out):
        recs, counts = PL.RecordExchange.unpack(out)
        assert counts.max() <= 16
        o = 0
        for r in range(world):
            got[r].extend((int(x[0]), int(x[1]) | (int(x[2]) << 8)) for x in recs[o:o + int(counts[r])])
            o += int(counts[r])
    prev = None
    for rnd in range(6):
        n = int(sizes[rnd, rank])
        rec = torch.zeros((n, REC_BYTES), dtype=torch.uint8)
        for i in range(n):
            rec[i, 0], rec[i, 1], rec[i, 2] = rank, serial & 255, serial >> 8
            serial += 1
        sent.append(n)
        out = ex.exchange(rec)
        if prev is not None:
            take(prev)          # a block stays valid until the exchange after the next: looked at one round late on purpose
        prev = out
    take(prev)
    for blk in ex.flush():
        take(blk)
    want = [[(r, i) for i in range(int(sizes[:, r].sum()))] for r in range(world)]
    q.put((rank, cpus, slowest, tot, pr, ok, ex.pending(), got == want, ex.exchanges))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world8_bench_plumbing_and_fixed_capacity_exchange():
    """SURVEY §8(e) at the size the driver will run it (8 ranks), on the CPU: the helpers bench.py times its N > 1 runs with
    (max-over-ranks time, summed counters, per-rank list), the record-exchange self check, per-rank CPU pinning, and the
    fixed-capacity RecordExchange with carry-over (no host agreement on sizes, same number of collectives on every rank)."""
    world = 8
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_plumbing_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, cpus, slowest, tot, pr, ok, pending, complete, nex in res:
        assert slowest == 1.0 + 0.25 * (world - 1)
        assert tot == [10.0 * world * (world + 1) / 2, 2.5 * world]
        assert pr == [100.0 + r for r in range(world)]
        assert ok is True, ok
        assert pending == 0 and complete and nex == res[0][8] and nex >= 7   # the same number of collectives on every rank
        assert cpus is None or len(cpus) >= 1
    pinned = [tuple(r[1]) for r in res if r[1] is not None]
    if len(pinned) == world and len(set(sum(pinned, ()))) >= world:   # enough CPUs here: the ranks' sets are disjoint
        assert len(set(sum(pinned, ()))) == sum(len(c) for c in pinned)
