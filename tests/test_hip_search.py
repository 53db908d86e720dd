"""GPU parity of the lock-step search (K4-K7): whole trees bit-identical to the UNMODIFIED reference
(golden mcts.json: root children, whole-tree digests, order of evaluated positions) and to the C
oracle on larger seeded batches."""
import numpy as np
import pytest
import torch

import fakenet
import searchdrive

pytestmark = pytest.mark.gpu


class _HipEngine:
    """Adapter: HIP SearchEngine with the host-array interface searchdrive expects."""

    def __init__(self, G, cap=200000):
        from cchess_zero_amd.engine import SearchEngine
        self.e = SearchEngine(G, cap)

    def reset(self, boards, side, rr):
        self.e.reset(boards, side, rr)

    def select(self, mode, mask=None):
        planes, need = self.e.select(mode, None if mask is None else mask.astype(np.uint8))
        return planes.cpu().numpy(), need.cpu().numpy()

    def expand_backup(self, logits, value):
        self.e.expand_backup(torch.from_numpy(logits).cuda(), torch.from_numpy(value).cuda())

    def root_stats(self):
        return self.e.root_stats_host()

    def advance(self, played):
        self.e.advance(played)

    def tree_dump(self, g):
        return self.e.tree_dump(g)

    def status(self):
        return self.e.status()[0].cpu().numpy()


def test_search_matches_reference_golden(mcts_golden):
    cases = mcts_golden["cases"]
    eng = _HipEngine(len(cases))
    results, logs = searchdrive.run_cases(eng, cases)
    assert not np.any(eng.status() & ~8)
    searchdrive.check_against_golden(results, logs, cases)


def test_search_matches_reference_at_depth(mcts_deep_golden):
    """The device search against the unmodified reference at the METRIC's depth (playout 1600: 58-75 k nodes per tree; a
    black-to-move root, a root at restrict_round 52, a second ply on the kept subtree) and on searches whose selected
    paths are 33-62 levels long — longer than CZ_PATH_MAX = 32, so both the recorded-path backup and the parent-pointer
    backup of k_expand_backup run (main.py:189-194,426-435), net leaves and the 60-ply draws at level 60 alike."""
    cases = mcts_deep_golden["cases"]
    eng = _HipEngine(len(cases))
    results, logs = searchdrive.run_cases(eng, cases)
    assert not np.any(eng.status() & ~8)
    searchdrive.check_against_golden(results, logs, cases)
    assert max(r["max_level"] for res in results for r in res) >= 60


@pytest.mark.parametrize("mode,advance", [("pos", "lds"), ("signed", "lds"), ("pos", "global")])
def test_search_vs_oracle_batch(rules_golden, mode, advance):
    """256 trees from corpus positions, 3 plies x 48 playouts, device-resident loop; compared with the
    oracle: needs_eval + planes every step, root stats and whole-tree dumps after each ply.  advance: the in-place
    compaction of cz_search_advance with the bitmap in LDS (default) or in global memory (pools too large for LDS)."""
    from oracle import oracle as O
    g = rules_golden
    ok = [(g["boards"][i] == 1).any() and (g["boards"][i] == 8).any() and g["counts"][i] > 0 for i in range(len(g["boards"]))]
    idx = np.nonzero(ok)[0][::11][:256]
    G = len(idx)
    boards, side = g["boards"][idx], g["side"][idx]
    rr = (np.arange(G) * 7 % 61).astype(np.int32)
    rr[::5] = 57
    hip = _HipEngine(G, 20000)
    if advance == "global":
        from cchess_zero_amd._lib import check, lib
        check(lib().cz_search_debug_advance_in_global_memory(hip.e.ctx.h, 1), "cz_search_debug_advance_in_global_memory")
    orc = O.Search(G, 20000)
    hip.reset(boards, side, rr)
    orc.reset(boards, side, rr)
    fwd = fakenet.make_forward(mode, 99)
    for ply in range(3):
        for step in range(49):
            m = 0 if step == 0 else 1
            hp, hn = hip.select(m)
            op, on = orc.select(m)
            assert np.array_equal(hn, on), (ply, step)
            assert np.array_equal(hp, op), (ply, step)
            logits, value = fwd(op)
            hip.expand_backup(logits, value)
            orc.expand_backup(logits, value)
        hs, os_ = hip.root_stats(), orc.root_stats()
        for k in ("label", "N", "count"):
            assert np.array_equal(hs[k], os_[k]), (ply, k)
        for k in ("Q", "P", "W"):
            assert np.array_equal(hs[k].view(np.uint32), os_[k].view(np.uint32)), (ply, k)
        for t in range(0, G, 17):
            assert np.array_equal(hip.tree_dump(t), orc.tree_dump(t)), (ply, t)
        hst, ost = hip.status(), orc.status()[0]
        assert np.array_equal(hst, ost)
        # play the most visited child (first max), park finished games
        played = np.full(G, 0xFFFF, np.uint16)
        for t in range(G):
            n = int(os_["count"][t])
            if n:
                played[t] = os_["label"][t, int(np.argmax(os_["N"][t, :n]))]
        hip.advance(played)
        orc.advance(played)
        hb, hsd, hrr = (x.cpu().numpy() for x in hip.e.root_state())
        ob, osd, orr = orc.root_state()
        assert np.array_equal(hb, ob) and np.array_equal(hsd, osd) and np.array_equal(hrr, orr)


def test_pool_exhaustion_is_reported():
    from oracle import oracle as O
    hip = _HipEngine(2, 64)
    b = np.tile(O.fen_to_board(O.START_FEN), (2, 1))
    hip.reset(b, np.zeros(2, np.uint8), np.zeros(2, np.int32))
    fwd = fakenet.make_forward("pos", 1)
    for step in range(4):
        p, n = hip.select(0 if step == 0 else 1)
        lg, v = fwd(p)
        hip.expand_backup(lg, v)
    assert np.all(hip.status() & 1)  # CZ_ST_POOL_EXHAUSTED, trees parked, no crash


_fc_logits_restated = searchdrive.fc_logits_restated


def test_expand_with_folded_policy_fc_vs_oracle(rules_golden):
    """cz_search_expand_backup_fc (policy FC evaluated inside the expansion, legal moves only) against the oracle
    fed with the full logits of the same FC restated in NumPy float32: whole trees bit-identical."""
    from oracle import oracle as O
    g = rules_golden
    ok = [(g["boards"][i] == 1).any() and (g["boards"][i] == 8).any() and g["counts"][i] > 0 for i in range(len(g["boards"]))]
    idx = np.nonzero(ok)[0][::13][:192]
    G = len(idx)
    boards, side = g["boards"][idx], g["side"][idx]
    rr = (np.arange(G) * 5 % 50).astype(np.int32)
    hip = _HipEngine(G, 20000)
    orc = O.Search(G, 20000)
    hip.reset(boards, side, rr)
    orc.reset(boards, side, rr)
    rng = np.random.default_rng(7)
    w = (rng.standard_normal((2086, 180)) * 0.3).astype(np.float32)
    b = (rng.standard_normal(2086) * 0.2).astype(np.float32)
    wd, bd = torch.from_numpy(w).cuda(), torch.from_numpy(b).cuda()
    for ply in range(2):
        for step in range(41):
            m = 0 if step == 0 else 1
            hp, hn = hip.select(m)
            op, on = orc.select(m)
            assert np.array_equal(hn, on) and np.array_equal(hp, op), (ply, step)
            # a "net": z and value are deterministic functions of the leaf planes
            feat = op.reshape(G, -1).astype(np.float32)
            z = np.maximum(0, (feat[:, :270] * 0.7 + feat[:, 270:540] * 0.31 + feat[:, 540:810] * 0.05 - 0.2)).astype(np.float32).reshape(G, 90, 3)
            z = (z + (feat.sum(axis=1, keepdims=True) % 7).astype(np.float32)[:, :, None] * 0.01).astype(np.float32)
            value = np.tanh(feat[:, ::9].sum(axis=1, keepdims=True) * 0.01 - 1.0).astype(np.float32)
            hip.e.expand_backup_fc(torch.from_numpy(z).cuda(), torch.from_numpy(value).cuda(), wd, bd)
            orc.expand_backup(_fc_logits_restated(z, w, b), value)
        hs, os_ = hip.root_stats(), orc.root_stats()
        for k in ("label", "N", "count"):
            assert np.array_equal(hs[k], os_[k]), (ply, k)
        for k in ("Q", "P", "W"):
            assert np.array_equal(hs[k].view(np.uint32), os_[k].view(np.uint32)), (ply, k)
        for t in range(0, G, 11):
            assert np.array_equal(hip.tree_dump(t), orc.tree_dump(t)), (ply, t)
        played = np.full(G, 0xFFFF, np.uint16)
        for t in range(G):
            n = int(os_["count"][t])
            if n:
                played[t] = os_["label"][t, int(np.argmax(os_["N"][t, :n]))]
        hip.advance(played)
        orc.advance(played)


def test_compact_batches_give_identical_trees(rules_golden):
    """SearchEngine.step with compact evaluation batches (terminal / drawn / parked trees get no net row, rows handed out
    by an atomic counter, net launches bounded by the device-side row count) against the same search with one row per
    tree: identical trees, with the real fused net.  Roots near the 60-ply limit make many leaves terminal."""
    from cchess_zero_amd.engine import SearchEngine
    from cchess_zero_amd.net import PolicyValueNet
    g = rules_golden
    ok = [(g["boards"][i] == 1).any() and (g["boards"][i] == 8).any() and g["counts"][i] > 0 for i in range(len(g["boards"]))]
    idx = np.nonzero(ok)[0][::9][:301]          # not a multiple of 4: the last workgroup of the trunk is ragged
    G = len(idx)
    boards, side = g["boards"][idx], g["side"][idx]
    rr = (np.arange(G) * 3 % 62).astype(np.int32)  # some trees start at rr 58..61: every leaf a draw
    net = PolicyValueNet(2, "cuda:0", torch.bfloat16, seed=4)
    assert net.fused_search
    engs = []
    for compact in (False, True):
        e = SearchEngine(G, 6000, plane_dtype=torch.bfloat16, channels=16)
        e.compact = compact
        e.reset(boards, side, rr)
        active = np.ones(G, np.uint8)
        active[5::17] = 0                          # parked trees
        e.search(net.forward_device, 40, active=active)
        engs.append(e)
    a, b = engs[0].root_stats_host(), engs[1].root_stats_host()
    for k in ("label", "N", "count"):
        assert np.array_equal(a[k], b[k]), k
    for k in ("Q", "P", "W"):
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k
    for t in range(0, G, 7):
        assert np.array_equal(engs[0].tree_dump(t), engs[1].tree_dump(t)), t
    rows, steps = engs[1].eval_totals()
    assert steps == 41 and 0 < rows < 41 * G      # fewer rows than trees x steps: some leaves needed no evaluation
    print("compact batches: %d rows for %d tree-steps (%.1f %%)" % (rows, 41 * G, 100.0 * rows / (41 * G)))


def _run_to_target(eng, fwds, playouts, extra):
    """Lock-steps with per-row fake nets until every tree has counted `playouts` simulations; returns the step count."""
    G = eng.e.G
    eng.e.set_terminal_extra(extra)
    eng.e.set_sim_target(playouts)
    steps = 0
    try:
        # root expansion first (MCTS_tree.main, main.py:475-487), like SearchEngine.search
        for mode in [0] + [1] * (playouts + 2):
            if mode == 1 and np.all(eng.e.status()[2].cpu().numpy() >= playouts):
                break
            planes, need = eng.select(mode)
            logits = np.zeros((G, 2086), np.float32)
            value = np.zeros((G, 1), np.float32)
            for g in range(G):
                if need[g]:
                    lg, v = fwds[g](planes[g])
                    logits[g], value[g] = lg[0], v[0]
            eng.expand_backup(logits, value)
            steps += mode
    finally:
        eng.e.set_sim_target(0)
        eng.e.set_terminal_extra(0)
    return steps


def _check_terminal_extra_first_ply(cases, extra):
    """First ply of every case with `extra` in-select completions per launch: golden root children, tree digest, number
    and order of net evaluations.  Returns the lock-steps saved."""
    import hashlib
    by_playouts = {}
    for c in cases:
        by_playouts.setdefault(c["plies"][0]["playouts"], []).append(c)
    saved = 0
    for playouts, group in sorted(by_playouts.items()):
        G = len(group)
        logs = [[] for _ in range(G)]
        fwds = [fakenet.make_forward(c["mode"], c["salt"], logs[i]) for i, c in enumerate(group)]
        boards = np.stack([searchdrive.fen_to_board(c["plies"][0]["state"]) for c in group])
        side = np.array([1 if c["plies"][0]["player"] == "b" else 0 for c in group], np.uint8)
        rr = np.array([c["plies"][0]["rr"] for c in group], np.int32)
        eng = _HipEngine(G)
        eng.reset(boards, side, rr)
        steps = _run_to_target(eng, fwds, playouts, extra)
        assert steps <= playouts
        saved += playouts - steps
        st = eng.root_stats()
        sims = eng.e.status()[2].cpu().numpy()
        assert np.all(sims == playouts)
        for g, c in enumerate(group):
            gp = c["plies"][0]
            n = int(st["count"][g])
            root = [(int(st["label"][g, i]), int(st["N"][g, i]), int(st["W"][g, i].view(np.uint32)),
                     int(st["Q"][g, i].view(np.uint32)), int(st["P"][g, i].view(np.uint32))) for i in range(n)]
            assert root == [tuple(x) for x in gp["root"]], c["name"]
            rec = eng.tree_dump(g)
            assert len(rec) == gp["tree_records"] and searchdrive.tree_digest(rec) == gp["tree_sha256"], c["name"]
            assert len(logs[g]) == gp["evals"], c["name"]
            if "eval_keys" in c:
                assert ["%016x" % k for k in logs[g]] == c["eval_keys"][:gp["evals"]], c["name"]
            elif len(c["plies"]) == 1:
                assert hashlib.sha256(",".join("%016x" % k for k in logs[g]).encode()).hexdigest() == c["eval_keys_sha256"], c["name"]
    return saved


@pytest.mark.parametrize("extra", [1, 3])
def test_terminal_simulations_complete_inside_select_golden(mcts_golden, extra):
    """cz_search_set_terminal_extra: simulations that end on a king capture / the 60-ply rule are backed up inside the select
    launch and the tree goes on to its next descent.  The trees must still be the UNMODIFIED reference's (golden root
    children, whole-tree digests, number and order of net evaluations) — with fewer lock-steps than playouts wherever such
    simulations occur."""
    saved = _check_terminal_extra_first_ply(mcts_golden["cases"], extra)
    print("terminal_extra=%d: %d lock-steps saved over the golden cases" % (extra, saved))
    assert saved > 0


def test_terminal_simulations_complete_inside_select_at_depth(mcts_deep_golden):
    """The same at playout 1600 and on the 33-62-level lines: a terminal leaf deeper than CZ_PATH_MAX is NOT completed inside
    the select launch (its path is not recorded) and goes through k_expand_backup's parent-pointer walk instead; the
    trees stay the reference's either way."""
    saved = _check_terminal_extra_first_ply(mcts_deep_golden["cases"], 4)
    print("terminal_extra=4 at depth: %d lock-steps saved" % saved)
    assert saved > 0


def test_terminal_extra_vs_oracle_batch(rules_golden):
    """256 trees, many of them near the 60-ply limit (every non-capturing leaf a draw): the device search with terminal
    simulations completing inside select against the oracle's plain one-simulation-per-step schedule — identical trees
    after the same number of simulations, in fewer lock-steps."""
    from oracle import oracle as O
    g = rules_golden
    ok = [(g["boards"][i] == 1).any() and (g["boards"][i] == 8).any() and g["counts"][i] > 0 for i in range(len(g["boards"]))]
    idx = np.nonzero(ok)[0][::11][:256]
    G, playouts = len(idx), 60
    boards, side = g["boards"][idx], g["side"][idx]
    rr = (np.arange(G) * 7 % 61).astype(np.int32)
    rr[::3] = 58
    hip = _HipEngine(G, 20000)
    orc = O.Search(G, 20000)
    hip.reset(boards, side, rr)
    orc.reset(boards, side, rr)
    fwd = fakenet.make_forward("signed", 31)
    for step in range(playouts + 1):
        op, on = orc.select(0 if step == 0 else 1)
        lg, v = fwd(op)
        orc.expand_backup(lg, v)
    hip.e.set_terminal_extra(2)
    hip.e.set_sim_target(playouts)
    busy_steps = 0   # lock-steps x trees still searching: G * playouts with one simulation per tree and step
    for mode in [0] + [1] * playouts:
        busy = (hip.e.status()[2].cpu().numpy() < playouts) & (hip.status() & ~8 == 0)
        if mode == 1 and not busy.any():
            break
        hp, hn = hip.select(mode)
        lg, v = fwd(hp)
        hip.expand_backup(lg, v)
        busy_steps += mode * int(busy.sum())
    hip.e.set_sim_target(0)
    hip.e.set_terminal_extra(0)
    assert busy_steps < 0.9 * G * playouts
    hs, os_ = hip.root_stats(), orc.root_stats()
    for k in ("label", "N", "count"):
        assert np.array_equal(hs[k], os_[k]), k
    for k in ("Q", "P", "W"):
        assert np.array_equal(hs[k].view(np.uint32), os_[k].view(np.uint32)), k
    assert np.array_equal(hip.status(), orc.status()[0])
    assert np.array_equal(hip.e.status()[2].cpu().numpy(), orc.status()[2])
    for t in range(0, G, 9):
        assert np.array_equal(hip.tree_dump(t), orc.tree_dump(t)), t
    print("terminal_extra=2 on %d trees: %d tree-steps instead of %d" % (G, busy_steps, G * playouts))


def test_reload_restarts_only_the_chosen_trees(rules_golden):
    """cz_search_reload (MCTS_tree.reload for the games that are over): the chosen trees become fresh roots at the given
    positions — their next search equals a fresh oracle search — and every other tree keeps every node."""
    from oracle import oracle as O
    g = rules_golden
    ok = [(g["boards"][i] == 1).any() and (g["boards"][i] == 8).any() and g["counts"][i] > 0 for i in range(len(g["boards"]))]
    idx = np.nonzero(ok)[0][::37][:24]
    G = len(idx)
    boards, side = g["boards"][idx], g["side"][idx]
    rr = (np.arange(G) * 5 % 50).astype(np.int32)
    fwd = fakenet.make_forward("pos", 7)

    def search(e, n):
        for step in range(n + 1):
            p, _ = e.select(0 if step == 0 else 1)
            lg, v = fwd(p)
            e.expand_backup(lg, v)

    hip = _HipEngine(G, 8000)
    hip.reset(boards, side, rr)
    search(hip, 20)
    before = [hip.tree_dump(t) for t in range(G)]
    which = np.zeros(G, np.uint8)
    which[[1, 4, 5, 17]] = 1
    nb, ns, nr = np.roll(boards, 3, axis=0), np.roll(side, 3), np.roll(rr, 3)
    hip.e.reload(which, nb, ns, nr)
    sims = hip.e.status()[2].cpu().numpy()
    assert np.array_equal(sims, np.where(which, 0, 20))
    for t in range(G):
        if not which[t]:
            assert np.array_equal(hip.tree_dump(t), before[t]), t
    rb, rs, rrr = (x.cpu().numpy() for x in hip.e.root_state())
    assert np.array_equal(rb, np.where(which[:, None] != 0, nb, boards)) and np.array_equal(rs, np.where(which, ns, side))
    assert np.array_equal(rrr, np.where(which, nr, rr))
    # the reloaded trees search like fresh ones; the others go on from where they were (30 simulations with the oracle)
    mask = which.astype(bool)
    for step in range(13):
        p, _ = hip.select(0 if step == 0 else 1, mask=mask)
        lg, v = fwd(p)
        hip.expand_backup(lg, v)
    orc = O.Search(G, 8000)
    orc.reset(nb, ns, nr)
    search(orc, 12)
    for t in np.nonzero(which)[0]:
        assert np.array_equal(hip.tree_dump(int(t)), orc.tree_dump(int(t))), t


def _dedup(seq):
    seen, out = set(), []
    for k in seq:
        if k not in seen:
            seen.add(k)
            out.append(k)
    return out


def test_eval_cache_keeps_the_golden_trees(mcts_golden):
    """cz_search_set_eval_cache: a leaf whose position the tree has evaluated before is expanded from the remembered node
    (children's labels and priors, backed-up value) inside the select launch.  The trees must still be the unmodified
    reference's — root children, whole-tree digests — and the net must see the golden sequence of positions with the
    repeats it was spared removed (a repeat may still reach the net when the launch's budget is used up)."""
    cases = mcts_golden["cases"]
    by_playouts = {}
    for c in cases:
        by_playouts.setdefault(c["plies"][0]["playouts"], []).append(c)
    spared = 0
    for playouts, group in sorted(by_playouts.items()):
        G = len(group)
        logs = [[] for _ in range(G)]
        fwds = [fakenet.make_forward(c["mode"], c["salt"], logs[i]) for i, c in enumerate(group)]
        boards = np.stack([searchdrive.fen_to_board(c["plies"][0]["state"]) for c in group])
        side = np.array([1 if c["plies"][0]["player"] == "b" else 0 for c in group], np.uint8)
        rr = np.array([c["plies"][0]["rr"] for c in group], np.int32)
        eng = _HipEngine(G)
        eng.e.set_eval_cache(True)
        eng.reset(boards, side, rr)
        _run_to_target(eng, fwds, playouts, 4)
        st = eng.root_stats()
        assert np.all(eng.e.status()[2].cpu().numpy() == playouts) and not np.any(eng.status() & ~8)
        for g, c in enumerate(group):
            gp = c["plies"][0]
            n = int(st["count"][g])
            root = [(int(st["label"][g, i]), int(st["N"][g, i]), int(st["W"][g, i].view(np.uint32)),
                     int(st["Q"][g, i].view(np.uint32)), int(st["P"][g, i].view(np.uint32))) for i in range(n)]
            assert root == [tuple(x) for x in gp["root"]], c["name"]
            rec = eng.tree_dump(g)
            assert len(rec) == gp["tree_records"] and searchdrive.tree_digest(rec) == gp["tree_sha256"], c["name"]
            gold = c["eval_keys"][:gp["evals"]]
            got = ["%016x" % k for k in logs[g]]
            assert len(got) <= len(gold) and _dedup(got) == _dedup(gold), c["name"]
            spared += len(gold) - len(got)
        hits, lookups = eng.e.eval_cache_stats()
        assert lookups >= hits >= 0
        eng.e.set_eval_cache(False)
    print("evaluation cache: %d net evaluations spared over the golden cases" % spared)
    assert spared > 0


@pytest.mark.parametrize("key_bits,extra", [(64, 4), (64, 0), (11, 4)])
def test_eval_cache_follows_the_tree_through_advances(rules_golden, key_bits, extra):
    """48 trees, 3 plies x 500 playouts with re-rooting in between, evaluation cache + terminal simulations inside select,
    against the oracle's plain schedule: identical root statistics and whole trees after every ply (the cache entries of
    kept nodes are remapped by cz_search_advance, the others dropped), with fewer net rows.
    extra = 0: the cache has its own per-launch budget and must hit with terminal_extra = 0 too.
    key_bits = 11: keys narrowed to 2048 values so that different positions DO share a key — every such match must be
    refused by the position check (collisions counted, taken as misses) and the trees must still be the oracle's."""
    from oracle import oracle as O
    g = rules_golden
    ok = [(g["boards"][i] == 1).any() and (g["boards"][i] == 8).any() and g["counts"][i] > 0 for i in range(len(g["boards"]))]
    idx = np.nonzero(ok)[0][::61][:48]
    G, playouts = len(idx), 500
    boards, side = g["boards"][idx], g["side"][idx]
    rr = (np.arange(G) * 3 % 40).astype(np.int32)
    hip = _HipEngine(G, 80000)
    orc = O.Search(G, 80000)
    from cchess_zero_amd._lib import check, lib
    check(lib().cz_search_debug_eval_cache_key_bits(hip.e.ctx.h, key_bits), "cz_search_debug_eval_cache_key_bits")
    hip.e.set_eval_cache(True)
    hip.reset(boards, side, rr)
    orc.reset(boards, side, rr)
    fwd = fakenet.make_forward("signed", 5)
    rows_hip = rows_orc = 0
    hits_by_ply = []
    for ply in range(3):
        for step in range(playouts + 1):
            op, on = orc.select(0 if step == 0 else 1)
            rows_orc += int(on.sum())
            lg, v = fwd(op)
            orc.expand_backup(lg, v)
        hip.e.set_terminal_extra(extra)
        hip.e.set_sim_target(playouts)
        for mode in [0] + [1] * playouts:
            busy = (hip.e.status()[2].cpu().numpy() < playouts) & (hip.status() & ~8 == 0)
            if mode == 1 and not busy.any():
                break
            hp, hn = hip.select(mode)
            rows_hip += int(hn.sum())
            lg, v = fwd(hp)
            hip.expand_backup(lg, v)
        hip.e.set_sim_target(0)
        hip.e.set_terminal_extra(0)
        hs, os_ = hip.root_stats(), orc.root_stats()
        for k in ("label", "N", "count"):
            assert np.array_equal(hs[k], os_[k]), (ply, k)
        for k in ("Q", "P", "W"):
            assert np.array_equal(hs[k].view(np.uint32), os_[k].view(np.uint32)), (ply, k)
        assert np.array_equal(hip.status(), orc.status()[0]), ply
        for t in range(0, G, 5):
            assert np.array_equal(hip.tree_dump(t), orc.tree_dump(t)), (ply, t)
        hits_by_ply.append(hip.e.eval_cache_stats()[0])
        # play the most visited move (ties: first), like get_action with temperature -> 0
        n = hs["N"].astype(np.int64).copy()
        n[np.arange(128)[None, :] >= hs["count"].astype(np.int64)[:, None]] = -1
        played = hs["label"][np.arange(G), n.argmax(axis=1)].astype(np.uint16)
        played[hs["count"] == 0] = 0xFFFF
        hip.advance(played)
        orc.advance(played)
    hits, lookups = hip.e.eval_cache_stats()
    collisions = hip.e.eval_cache_collisions()
    print("evaluation cache (%d-bit keys, terminal_extra %d) over 3 plies x %d playouts x %d trees: %d hits / %d lookups (by ply: %s), "
          "%d key collisions refused, net rows %d instead of %d" %
          (key_bits, extra, playouts, G, hits, lookups, np.diff([0] + hits_by_ply).tolist(), collisions, rows_hip, rows_orc))
    assert hits > 0 and rows_hip < rows_orc and hits_by_ply[2] > hits_by_ply[1] > hits_by_ply[0]
    assert (collisions > 100) if key_bits < 64 else (collisions == 0)
    hip.e.set_eval_cache(False)
    check(lib().cz_search_debug_eval_cache_key_bits(hip.e.ctx.h, 64), "cz_search_debug_eval_cache_key_bits")


@pytest.mark.parametrize("key_bits", [64, 11])
def test_xcache_lends_evaluations_across_trees_and_keeps_every_tree(key_bits):
    """The cross-tree level of the evaluation cache (cz_search_set_xcache): 61 trees rooted at the start position, at its 44
    successors and at 16 second-ply positions — the root of one tree is an inner node of another, as the openings of self-play
    games are — 2 plies x 300 playouts with a re-root in between, against the oracle's plain schedule: identical root
    statistics and whole trees; evaluations ARE lent across trees (hits > 0, fewer net rows than with the per-tree level
    alone); with keys narrowed to 2 048 values every false match is refused by the stored position."""
    from oracle import oracle as O
    from cchess_zero_amd._lib import check, lib
    b0 = O.fen_to_board(O.START_FEN)
    boards, side = [b0], [0]
    for m in O.legal_moves(b0, 0):
        boards.append(O.apply_move(b0, int(m))[0]); side.append(1)
    b1 = boards[20]
    for m in O.legal_moves(b1, 1)[:16]:
        boards.append(O.apply_move(b1, int(m))[0]); side.append(0)
    boards, side = np.stack(boards), np.array(side, np.uint8)
    G, playouts = len(boards), 300
    rr = np.zeros(G, np.int32)
    fwd = fakenet.make_forward("signed", 5)

    def run(xc):
        hip = _HipEngine(G, 60000)
        check(lib().cz_search_debug_eval_cache_key_bits(hip.e.ctx.h, key_bits), "cz_search_debug_eval_cache_key_bits")
        hip.e.set_eval_cache(True)
        if xc:
            hip.e.set_xcache(12)
        hip.reset(boards, side, rr)
        rows, dumps = 0, []
        for ply in range(2):
            hip.e.set_terminal_extra(4)
            hip.e.set_sim_target(playouts)
            for mode in [0] + [1] * playouts:
                busy = (hip.e.status()[2].cpu().numpy() < playouts) & (hip.status() & ~8 == 0)
                if mode == 1 and not busy.any():
                    break
                hp, hn = hip.select(mode)
                rows += int(hn.sum())
                lg, v = fwd(hp)
                hip.expand_backup(lg, v)
            hip.e.set_sim_target(0)
            hip.e.set_terminal_extra(0)
            hs = hip.root_stats()
            dumps.append((hs, [hip.tree_dump(t) for t in range(0, G, 4)], hip.status().copy()))
            n = hs["N"].astype(np.int64).copy()
            n[np.arange(128)[None, :] >= hs["count"].astype(np.int64)[:, None]] = -1
            played = hs["label"][np.arange(G), n.argmax(axis=1)].astype(np.uint16)
            played[hs["count"] == 0] = 0xFFFF
            hip.advance(played)
        st = hip.e.xcache_stats() if xc else None
        hip.e.set_eval_cache(False)
        check(lib().cz_search_debug_eval_cache_key_bits(hip.e.ctx.h, 64), "cz_search_debug_eval_cache_key_bits")
        return rows, dumps, st

    orc = O.Search(G, 60000)
    orc.reset(boards, side, rr)
    odumps = []
    for ply in range(2):
        for step in range(playouts + 1):
            op, on = orc.select(0 if step == 0 else 1)
            lg, v = fwd(op)
            orc.expand_backup(lg, v)
        os_ = orc.root_stats()
        odumps.append((os_, [orc.tree_dump(t) for t in range(0, G, 4)], orc.status()[0].copy()))
        n = os_["N"].astype(np.int64).copy()
        n[np.arange(128)[None, :] >= os_["count"].astype(np.int64)[:, None]] = -1
        played = os_["label"][np.arange(G), n.argmax(axis=1)].astype(np.uint16)
        played[os_["count"] == 0] = 0xFFFF
        orc.advance(played)
    rows_tree, d_tree, _ = run(False)
    rows_xc, d_xc, st = run(True)
    for got in (d_tree, d_xc):
        for ply in range(2):
            hs, dumps, status = got[ply]
            os_, odump, ostatus = odumps[ply]
            for k in ("label", "N", "count"):
                assert np.array_equal(hs[k], os_[k]), (ply, k)
            for k in ("Q", "P", "W"):
                assert np.array_equal(hs[k].view(np.uint32), os_[k].view(np.uint32)), (ply, k)
            assert np.array_equal(status, ostatus)
            for a, b in zip(dumps, odump):
                assert np.array_equal(a, b), ply
    print("cross-tree cache (%d-bit keys), %d trees x 2 plies x %d playouts: net rows %d with the per-tree level alone, %d with the "
          "cross-tree level; %s" % (key_bits, G, playouts, rows_tree, rows_xc, st))
    assert st["written"] > 0 and st["lookups"] >= st["hits"]
    if key_bits == 64:      # (narrowed keys all fall into ONE bucket of the cross-tree table: 64 entries; what counts there is
        assert st["hits"] > 0 and rows_xc < rows_tree     #  that every false match is refused — the identity asserted above)


def test_advance_ready_device_driver_matches_host_logic():
    """cz_search_pick_ready -> cz_search_advance -> cz_search_reload_finished (the greedy driver bench.py's search loop
    runs every few steps) against the same decisions taken on the host from cz_search_status / root_stats / root_state:
    per-tree thresholds, first-maximum most-visited child (get_action's temperature -> 0 limit, main.py:1332-1341),
    check_end + reload from the start position (main.py:1380-1392,255-258).  Two engines, the same net, 160 lock-steps
    with a check every 8: every root, counter, statistic, the banked simulations and the restart count must agree."""
    import bench
    from cchess_zero_amd.engine import SearchEngine
    from cchess_zero_amd.net import PolicyValueNet
    from cchess_zero_amd.rules import Rules
    G, playout = 512, 24
    boards, side, rr = bench.synth_positions(Rules(), G, 4242, max_ply=160)   # late positions: some games end within the run
    net = PolicyValueNet(2, "cuda:0", torch.float16, seed=5)
    dev = boards.device
    start_b = torch.from_numpy(np.tile(bench.START, (G, 1))).to(dev)
    start_s, start_r = torch.zeros(G, dtype=torch.uint8, device=dev), torch.zeros(G, dtype=torch.int32, device=dev)
    thr0 = torch.randint(4, 40, (G,), generator=torch.Generator(device=dev).manual_seed(1), device=dev, dtype=torch.int32)
    out = []
    for device_driver in (True, False):
        eng = SearchEngine(G, 8192, plane_dtype=torch.float16, channels=16)
        eng.reset(boards, side, rr)
        eng.set_terminal_extra(2)
        eng.set_sim_target(playout)
        thr = thr0.clone()
        banked = torch.zeros(1, dtype=torch.int64, device=dev)
        reloaded = torch.zeros(1, dtype=torch.int64, device=dev)
        eng.step(net.forward_device, mode=0)
        for step in range(160):
            eng.step(net.forward_device, mode=1)
            if (step + 1) % 8:
                continue
            if device_driver:
                eng.advance_ready(thr, playout, start_b, start_s, start_r, banked, reloaded)
                continue
            st, _, sims, _ = eng.status()
            ready = (sims >= thr) | ((st & 1) != 0)
            thr = torch.where(ready, torch.full_like(thr, playout), thr)
            banked += (sims.to(torch.int64) * ready).sum()
            rs = eng.root_stats()
            n = rs["N"].clone()
            cnt = (rs["count"].to(torch.int64) & 0xFFFF).unsqueeze(1)
            n[torch.arange(128, device=dev).unsqueeze(0) >= cnt] = -1
            played = rs["label"].gather(1, n.argmax(dim=1, keepdim=True)).squeeze(1)
            eng.advance(torch.where(ready & (cnt.squeeze(1) > 0), played, torch.full_like(played, -1)))
            b, _, r = eng.root_state()
            over = ready & (~(b == 1).any(dim=1) | ~(b == 8).any(dim=1) | (r >= 60) | (cnt.squeeze(1) == 0))
            eng.reload(over, start_b, start_s, start_r)
            reloaded += over.sum()
        rs = eng.root_stats()
        out.append(dict(thr=thr.clone(), banked=int(banked.item()), reloaded=int(reloaded.item()),
                        state=[x.clone() for x in eng.root_state()], status=[x.clone() for x in eng.status()],
                        stats={k: v.clone() for k, v in rs.items()}))
        eng.set_sim_target(0)
        eng.set_terminal_extra(0)
    a, b = out
    assert a["banked"] == b["banked"] > 0 and a["reloaded"] == b["reloaded"] > 0
    assert torch.equal(a["thr"], b["thr"])
    for x, y in zip(a["state"] + a["status"], b["state"] + b["status"]):
        assert torch.equal(x, y)
    for k in a["stats"]:
        x, y = a["stats"][k], b["stats"][k]
        assert torch.equal(x.view(torch.int32) if x.dtype.is_floating_point else x, y.view(torch.int32) if y.dtype.is_floating_point else y), k
    print("advance_ready: %d simulations banked, %d games restarted, identical under both drivers" % (a["banked"], a["reloaded"]))
