"""The device-resident self-play loop (cchess_zero_amd/selfplay.py over cz_selfplay_* / cz_search_advance) with the real
fused net: continuous re-seeding keeps every slot busy, the records are well-formed training tuples of the reference's
self-play (main.py:1493-1554), a tree that fills its node pool recovers at the next advance."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

START_FEN = "RNBAKABNR/9/1C5C1/P1P1P1P1P/9/9/p1p1p1p1p/1c5c1/9/rnbakabnr"


def _setup(G, cap, playouts, continuous=True, seed=3, **kw):
    from cchess_zero_amd.engine import SearchEngine
    from cchess_zero_amd.net import PolicyValueNet
    from cchess_zero_amd.selfplay import SelfPlay
    from oracle import oracle as O
    net = PolicyValueNet(2, "cuda:0", torch.float16, seed=2)
    eng = SearchEngine(G, cap, plane_dtype=torch.float16, channels=16)
    sp = SelfPlay(eng, net, playouts, exploration=True, temperature=1.0, seed=seed, continuous=continuous, **kw)
    sp.start(np.tile(O.fen_to_board(START_FEN), (G, 1)), np.zeros(G, np.uint8), np.zeros(G, np.int32))
    return net, eng, sp


def _check_records(rec, playouts):
    """Every record is a (state, visit policy, z) tuple of a legal self-play game; returns the number of games."""
    from cchess_zero_amd.selfplay import to_dense, unpack_records
    from oracle import oracle as O
    u = unpack_records(rec)
    n = len(rec)
    starts = list(np.nonzero(u["ply"] == 0)[0]) + [n]
    assert starts[0] == 0
    for a, b in zip(starts[:-1], starts[1:]):
        assert np.array_equal(u["ply"][a:b], np.arange(b - a)), "records of a game are contiguous and in ply order"
        assert np.array_equal(u["side"][a:b], np.arange(b - a) % 2), "red moves first, sides alternate"
        assert np.array_equal(u["boards"][a], O.fen_to_board(START_FEN))
        z = u["z"][a:b].astype(int)
        # one result per game, seen from the mover: constant up to the side
        res = z * np.where(u["side"][a:b] == 0, 1, -1)
        assert len(set(res.tolist())) == 1 and res[0] in (-1, 0, 1)
        board = u["boards"][a].copy()
        for j in range(a, b):
            k = int(u["counts"][j])
            mv = O.legal_moves(u["boards"][j], int(u["side"][j]))
            assert k == len(mv) and np.array_equal(u["labels"][j, :k], mv), "root children = legal moves, generation order"
            assert np.all(u["labels"][j, k:] == 0xFFFF) and np.all(u["visits"][j, k:] == 0)
            v = int(u["visits"][j, :k].astype(np.int64).sum())
            assert v >= playouts if j > a else v == playouts, (j - a, v)   # a fresh root has exactly `playouts` visits below it
            if j + 1 < b:   # the next recorded position follows from this one by one of its legal moves
                nxt = u["boards"][j + 1]
                diff = np.nonzero(nxt != u["boards"][j])[0]
                assert 1 <= len(diff) <= 2
        last = u["boards"][b - 1]
        assert (last == 1).any() and (last == 8).any(), "both kings are on the board before the last move"
    planes, pi, z = to_dense(rec, 1.0, exact=False)
    assert np.allclose(pi.sum(axis=1), 1.0, atol=1e-12) and planes.shape == (n, 9, 10, 14)
    return len(starts) - 1


def test_continuous_selfplay_keeps_every_slot_busy():
    G, playouts, plies = 96, 6, 260
    net, eng, sp = _setup(G, 4096, playouts)
    recs = []
    for chunk in range(plies // 20):
        sp.run(20)
        recs.append(sp.drain())
    st = sp.stats()
    rec = np.concatenate(recs, axis=0)
    print("continuous self-play: %d plies x %d slots: %d games finished (%d red, %d black, %d draws), %d records, %d stalled" %
          (plies, G, st["games"], st["red_wins"], st["black_wins"], st["draws"], st["plies"], st["stalled"]))
    # every slot searched every ply: the batch never decayed
    assert st["sims"] == plies * G * playouts
    assert st["stalled"] == 0 and st["dropped"] == 0 and st["games"] >= G // 2
    assert st["games"] == st["red_wins"] + st["black_wins"] + st["draws"] and st["plies"] == len(rec)
    assert not bool(eng.status()[0].any())
    games = _check_records(rec, playouts)
    assert games == st["games"]
    # games in progress keep their partial histories on the device: the plies recorded so far
    assert sp.active().all()


def test_parking_mode_plays_one_game_per_slot():
    G, playouts = 24, 4
    net, eng, sp = _setup(G, 4096, playouts, continuous=False, max_plies=400)
    rec = sp.play()
    st = sp.stats()
    assert st["games"] == G and not bool(sp.active().any())
    assert _check_records(rec, playouts) == G
    # parked slots cost the search nothing afterwards
    before = st["sims"]
    sp.step_ply()
    assert sp.stats()["sims"] == before


def test_full_node_pool_recovers_at_the_advance():
    """A pool too small for a ply's expansions: the tree stops expanding (CZ_ST_POOL_EXHAUSTED), its move is still chosen
    from the visits it has, and cz_search_advance — which compacts the kept subtree in place — clears the flag, so the
    game goes on (round 1 parked such a tree for the rest of its game and recorded junk plies)."""
    G, playouts = 16, 24
    net, eng, sp = _setup(G, 400, playouts)   # ~40 new nodes per simulation: 400 nodes fill up within a ply
    saw_full = 0
    for ply in range(30):
        eng.search(net.forward_device, playouts)
        st = eng.status()[0].cpu().numpy()
        saw_full += int((st & 1).sum())
        assert not np.any(st & ~1)
        # the rest of step_ply: choose / advance / adjudicate / flush (search again on top would double the playouts)
        sp.playouts = 0
        sp.step_ply()
        sp.playouts = playouts
        assert not np.any(eng.status()[0].cpu().numpy() & 1), "advance clears POOL_EXHAUSTED"
    assert saw_full > 0, "the pool was meant to overflow"
    s = sp.stats()
    assert s["stalled"] == 0
    rec = sp.drain()
    if len(rec):
        from cchess_zero_amd.selfplay import unpack_records
        u = unpack_records(rec)
        assert np.all(u["counts"] > 0) and np.all(u["visits"].astype(np.int64).sum(axis=1) > 0)


def test_asynchronous_plies_keep_records_well_formed():
    """SelfPlay.run_async: every game moves at its own pace (when its search has completed `playouts` simulations), terminal
    simulations complete inside the select launches.  The records must be the same kind of tuples as with lock-step plies:
    every root searched with exactly `playouts` simulations on top of the kept subtree."""
    G, playouts = 96, 6
    net, eng, sp = _setup(G, 4096, playouts)
    recs = []
    for chunk in range(12):
        sp.run_async(150, every=4, terminal_extra=2)
        recs.append(sp.drain())
    st = sp.stats()
    rec = np.concatenate(recs, axis=0)
    print("asynchronous plies: 1800 steps x %d slots: %d games finished, %d records, %d sims (%.3f per net row)" %
          (G, st["games"], st["plies"], st["sims"], st["sims"] / (1800.0 * G)))
    assert st["stalled"] == 0 and st["dropped"] == 0 and st["games"] >= G // 2
    assert st["games"] == st["red_wins"] + st["black_wins"] + st["draws"] and st["plies"] == len(rec)
    assert not bool(eng.status()[0].any())
    assert _check_records(rec, playouts) == st["games"]
    # simulations: every move played had its full search; a lock-step completes at least one simulation per tree that is
    # not waiting for its move, and more than one where terminal simulations completed inside select
    assert st["sims"] >= st["plies"] * playouts
    assert int(eng.status()[2].max().item()) <= playouts


def test_eval_cache_and_terminal_extra_leave_real_net_selfplay_unchanged():
    """The claim behind the evaluation cache, checked with the REAL fused net: a position evaluated twice gets the same
    priors and value bit for bit (every row of a batch is computed independently of the others), so whole self-play games
    — visit counts, sampled moves, results — are byte-identical with the cache and in-select terminal simulations on or off."""
    G, playouts, plies = 64, 24, 30

    def run(cache, xcache=0):
        net, eng, sp = _setup(G, 8192, playouts, seed=11)
        if cache:
            eng.set_eval_cache(True)
            eng.set_terminal_extra(4)
            if xcache:
                eng.set_xcache(xcache)
        sp.run(plies)
        rec = sp.drain()
        st = sp.stats()
        hits = eng.eval_cache_stats() if cache else (0, 0)
        xst = eng.xcache_stats() if xcache else None
        if cache:
            eng.set_terminal_extra(0)
            eng.set_eval_cache(False)
        return rec, st, hits, xst

    rec0, st0, _, _ = run(False)
    rec1, st1, hits, _ = run(True)
    rec2, st2, hits2, xst = run(True, 14)      # + the cross-tree level: all 64 games start from the same position
    # a table far too small for the run (2^8 entries = four buckets): full buckets replace their deepest-in-game entry by a
    # shallower position (round 6) — records still byte-identical, replacements and refusals both seen
    rec3, st3, _, xst3 = run(True, 8)
    print("cross-tree level with 256 entries:", xst3)
    assert rec0.shape == rec3.shape and np.array_equal(rec0, rec3)
    assert xst3["replaced"] > 0 and xst3["lost"] > 0 and xst3["written"] >= 256 + xst3["replaced"] - 8 and xst3["hits"] > 0
    assert xst["replaced"] == 0 or xst["written"] > xst["replaced"]
    print("real net, %d games x %d plies x %d playouts: %d records; cache hits %d / %d lookups; with the cross-tree level %s" %
          (G, plies, playouts, len(rec0), hits[0], hits[1], xst))
    assert len(rec0) > 0 and rec0.shape == rec1.shape and np.array_equal(rec0, rec1)
    assert rec0.shape == rec2.shape and np.array_equal(rec0, rec2)
    keys = ("games", "red_wins", "black_wins", "draws", "plies")
    assert {k: st0[k] for k in keys} == {k: st1[k] for k in keys} == {k: st2[k] for k in keys}
    assert hits[0] > 0 and xst["hits"] > 0 and xst["written"] > 0
    assert st2["lock_steps"] <= st1["lock_steps"] <= st0["lock_steps"]


def test_ring_overflow_is_reported_not_silently_drained():
    """a finished game whose records would overwrite undrained rows is dropped by cz_selfplay_flush while the
    cursor still advances — drain must not hand back the stale slots: it raises, and the statistics carry the count."""
    G, playouts = 64, 4
    net, eng, sp = _setup(G, 4096, playouts, ring_records=256)   # far too small for 64 continuously re-seeded games
    with pytest.raises(RuntimeError, match="ring overflow"):
        for _ in range(40):
            sp.run(20)          # never drained in between: the ring fills with finished games
            if sp.stats()["dropped"] > 0:
                sp.drain()
                break
        else:
            sp.drain()
    assert sp.stats()["dropped"] > 0 and sp.overflow_intervals == 1
    # the interval is gone, not handed out; the loop can go on ("skip" mode of a long training run returns no rows either)
    assert len(sp.drain()) == 0
    sp.run(4)
    rec = sp.drain_device(on_overflow="skip")
    assert sp.overflow_intervals in (1, 2) and (len(rec) == 0 or sp.overflow_intervals == 1)


def test_search_early_exit_ignores_inactive_trees_behind_a_raw_mask_pointer():
    """in parking mode (continuous=False) SelfPlay hands search() the active mask as a RAW DEVICE POINTER; the early
    exit of a terminal_extra > 0 search must look through it — inactive slots never reach the target and used to keep every
    search running for its full `playouts` lock-steps.  One active tree whose root can capture the king (kings facing on an open
    file: most simulations end inside the select launch) among seven inactive ones: the search ends as soon as THAT tree has
    its playouts, with the statistics of a plain search."""
    import ctypes as C
    from cchess_zero_amd.engine import SearchEngine
    from cchess_zero_amd.net import PolicyValueNet
    from oracle import oracle as O
    G, playouts = 8, 64
    b = np.stack([O.fen_to_board("4K4/9/9/9/9/9/9/9/9/4k4")] + [O.fen_to_board(START_FEN)] * (G - 1))
    side, rr = np.zeros(G, np.uint8), np.zeros(G, np.int32)
    net = PolicyValueNet(2, "cuda:0", torch.float16, seed=2)
    mask = torch.zeros(G, dtype=torch.uint8, device="cuda")
    mask[0] = 1
    out = []
    for extra in (0, 4):
        eng = SearchEngine(G, 8192, plane_dtype=torch.float16, channels=16)
        eng.reset(b, side, rr)
        eng.set_terminal_extra(extra)
        steps = eng.search(net.forward_device, playouts, active=C.c_void_p(mask.data_ptr()))
        eng.set_terminal_extra(0)
        st = eng.root_stats_host()
        sims = eng.status()[2].cpu().numpy()
        assert sims[0] == playouts and not sims[1:].any()
        out.append((steps, st["N"][0].copy(), st["W"][0].view(np.uint32).copy()))
    (s0, n0, w0), (s1, n1, w1) = out
    print("one active tree behind a raw mask pointer: %d lock-steps with terminal_extra 0, %d with 4" % (s0, s1))
    assert s0 == playouts and s1 < playouts and np.array_equal(n0, n1) and np.array_equal(w0, w1)
