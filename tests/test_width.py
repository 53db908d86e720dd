"""k simulations in flight per tree (virtual loss; the reference's search_threads).
 * k = 1 through the k-wide code path must reproduce the UNMODIFIED reference's trees (golden mcts.json);
 * k > 1: the HIP kernels must equal the C oracle's restatement of the same batched schedule bit for bit, and the
   trees must satisfy the virtual-loss-free invariants after every step."""
import numpy as np
import pytest

import fakenet
import searchdrive
from oracle import oracle as O


class _OracleK:
    """searchdrive adapter: one czo_search per tree, k-wide calls with k = 1."""

    def __init__(self, G, K=1, cap=200000):
        self.G, self.K = G, K
        self.e = [O.Search(1, cap) for _ in range(G)]

    def reset(self, boards, side, rr):
        for g in range(self.G):
            self.e[g].reset(boards[g:g + 1], side[g:g + 1], rr[g:g + 1])

    def select(self, mode, mask=None):
        planes = np.zeros((self.G, 9, 10, 14), np.float32)
        need = np.zeros(self.G, np.uint8)
        self._ran = np.zeros(self.G, bool)
        for g in range(self.G):
            if mask is not None and not mask[g]:
                continue
            p, n = self.e[g].select_k(mode, 1)
            planes[g], need[g] = p[0], n[0]
            self._ran[g] = True
        return planes, need

    def expand_backup(self, logits, value):
        for g in range(self.G):
            if self._ran[g]:
                self.e[g].expand_backup_k(1, logits[g:g + 1], value[g:g + 1])

    def root_stats(self):
        st = [e.root_stats() for e in self.e]
        return {k: np.concatenate([s[k] for s in st]) for k in st[0]}

    def advance(self, played):
        for g in range(self.G):
            self.e[g].advance(played[g:g + 1])

    def tree_dump(self, g):
        return self.e[g].tree_dump(0)


def test_oracle_width1_schedule_equals_reference(mcts_golden):
    cases = mcts_golden["cases"]
    eng = _OracleK(len(cases))
    results, logs = searchdrive.run_cases(eng, cases)
    searchdrive.check_against_golden(results, logs, cases)


def _invariants(rec, root_sum_expected=None):
    from test_scale_properties import _tree_invariants
    _tree_invariants(rec)
    if root_sum_expected is not None:
        assert int(sum(int(r[2]) for r in rec if int(r[0]) == 0)) == root_sum_expected


@pytest.mark.parametrize("K", [2, 4, 16])
def test_oracle_width_k_invariants(rules_golden, K):
    g = rules_golden
    ok = [(g["boards"][i] == 1).any() and (g["boards"][i] == 8).any() and g["counts"][i] > 0 for i in range(len(g["boards"]))]
    idx = np.nonzero(ok)[0][::97][:24]
    G = len(idx)
    s = O.Search(G, 60000)
    s.reset(g["boards"][idx], g["side"][idx], (np.arange(G) * 5 % 58).astype(np.int32))
    fwd = fakenet.make_forward("pos", 5)
    for step in range(40):
        planes, need = s.select_k(0 if step == 0 else 1, K)
        logits, value = fwd(planes)
        s.expand_backup_k(K, logits, value)
        if step in (1, 7, 39):
            st, nodes, sims = s.status()
            for t in range(0, G, 5):
                _invariants(s.tree_dump(t), int(sims[t]))   # no virtual loss left behind, every sim counted once
    st, nodes, sims = s.status()
    assert not np.any(st & ~8)
    assert np.all(sims <= 39 * K) and np.all(sims >= 39)   # at least one, at most K simulations per step


@pytest.mark.gpu
def test_hip_width1_path_equals_reference(mcts_golden):
    """the k-wide HIP kernels with k = 1 reproduce the reference trees bit for bit"""
    import torch
    from cchess_zero_amd import _lib
    from cchess_zero_amd.engine import SearchEngine
    import ctypes as C

    class Eng:
        def __init__(self, G):
            self.e = SearchEngine(G, 200000)
            self.G = G

        def reset(self, b, s, rr):
            self.e.reset(b, s, rr)

        def select(self, mode, mask=None):
            act = None if mask is None else torch.from_numpy(mask.astype(np.uint8)).cuda()
            _lib.check(_lib.lib().cz_search_select_k(self.e.ctx.h, int(mode), 1, C.c_void_p(act.data_ptr()) if act is not None else None,
                                                     C.c_void_p(self.e.planes.data_ptr()), _lib.F32, 14, C.c_void_p(self.e.need.data_ptr())), "select_k")
            return self.e.planes.cpu().numpy(), self.e.need.cpu().numpy()

        def expand_backup(self, logits, value):
            lg, v = torch.from_numpy(logits).cuda(), torch.from_numpy(value).cuda().contiguous()
            _lib.check(_lib.lib().cz_search_expand_backup_k(self.e.ctx.h, 1, C.c_void_p(lg.data_ptr()), C.c_void_p(v.data_ptr()), _lib.F32), "expand_k")

        def root_stats(self):
            return self.e.root_stats_host()

        def advance(self, played):
            self.e.advance(played)

        def tree_dump(self, g):
            return self.e.tree_dump(g)

    cases = mcts_golden["cases"]
    eng = Eng(len(cases))
    results, logs = searchdrive.run_cases(eng, cases)
    searchdrive.check_against_golden(results, logs, cases)


@pytest.mark.gpu
@pytest.mark.parametrize("K,mode", [(2, "pos"), (4, "signed"), (16, "pos")])
def test_hip_width_k_equals_oracle(rules_golden, K, mode):
    import torch
    from cchess_zero_amd.engine import SearchEngine
    g = rules_golden
    ok = [(g["boards"][i] == 1).any() and (g["boards"][i] == 8).any() and g["counts"][i] > 0 for i in range(len(g["boards"]))]
    idx = np.nonzero(ok)[0][::13][:192]
    G = len(idx)
    boards, side = g["boards"][idx], g["side"][idx]
    rr = (np.arange(G) * 7 % 61).astype(np.int32)
    rr[::5] = 57
    cap = 3000 + 2 * 25 * K * 90   # the oracle does not compact on advance: room for both plies
    hip = SearchEngine(G, cap, width=K)
    orc = O.Search(G, cap)
    hip.reset(boards, side, rr)
    orc.reset(boards, side, rr)
    fwd = fakenet.make_forward(mode, 77)
    for ply in range(2):
        for step in range(25):
            m = 0 if step == 0 else 1
            hp, hn = hip.select(m)
            op, on = orc.select_k(m, K)
            assert np.array_equal(hn.cpu().numpy(), on), (ply, step)
            assert np.array_equal(hp.cpu().numpy(), op), (ply, step)
            logits, value = fwd(op)
            hip.expand_backup(torch.from_numpy(logits).cuda(), torch.from_numpy(value).cuda())
            orc.expand_backup_k(K, logits, value)
        hs, os_ = hip.root_stats_host(), orc.root_stats()
        for k in ("label", "N", "count"):
            assert np.array_equal(hs[k], os_[k]), (ply, k)
        for k in ("Q", "P", "W"):
            assert np.array_equal(hs[k].view(np.uint32), os_[k].view(np.uint32)), (ply, k)
        hst, hnodes, hsims, _ = (x.cpu().numpy() for x in hip.status())
        ost, onodes, osims = orc.status()
        assert np.array_equal(hst, ost) and np.array_equal(hsims, osims)
        if ply == 0:   # k_advance compacts the kept subtree, the oracle keeps the whole pool
            assert np.array_equal(hnodes, onodes)
        for t in range(0, G, 23):
            rec = hip.tree_dump(t)
            assert np.array_equal(rec, orc.tree_dump(t)), (ply, t)
            _invariants(rec, int(hsims[t]) if ply == 0 else None)   # after a re-root the children keep earlier visits
        played = np.full(G, 0xFFFF, np.uint16)
        for t in range(G):
            n = int(os_["count"][t])
            if n:
                played[t] = os_["label"][t, int(np.argmax(os_["N"][t, :n]))]
        hip.advance(played)
        orc.advance(played)


def _budget_search_oracle(s, fwd, K, playouts):
    """SearchEngine.search's k > 1 schedule on the oracle: budgeted steps until every tree has its playouts."""
    planes, need = s.select_k(0, K)
    lg, v = fwd(planes)
    s.expand_backup_k(K, lg, v)
    base = s.status()[2].copy()
    assert np.all(base == base[0])
    target = int(base[0]) + playouts
    s.set_sim_target(target)
    steps = 0
    try:
        while steps < 5 * playouts + 8:
            st, _, sims = s.status()
            if steps >= (playouts + K - 1) // K and not np.any((sims < target) & ((st & ~8) == 0)):
                break
            planes, need = s.select_k(1, K)
            lg, v = fwd(planes)
            s.expand_backup_k(K, lg, v)
            steps += 1
    finally:
        s.set_sim_target(0)
    return steps


@pytest.mark.parametrize("K,playouts", [(4, 50), (16, 200), (16, 37)])
def test_oracle_width_k_budget_gives_exact_playouts(rules_golden, K, playouts):
    """with k simulations in flight a step can complete fewer than k (abandoned descents); the per-tree budget
    makes a search end with exactly `playouts` simulations per tree, like MCTS_tree.main (main.py:489-493)."""
    g = rules_golden
    ok = [(g["boards"][i] == 1).any() and (g["boards"][i] == 8).any() and g["counts"][i] > 0 for i in range(len(g["boards"]))]
    idx = np.nonzero(ok)[0][::61][:32]
    G = len(idx)
    s = O.Search(G, 60000)
    s.reset(g["boards"][idx], g["side"][idx], np.zeros(G, np.int32))
    steps = _budget_search_oracle(s, fakenet.make_forward("pos", 9), K, playouts)
    st, nodes, sims = s.status()
    assert np.all(sims == playouts), (sims.min(), sims.max())
    assert steps >= (playouts + K - 1) // K
    rs = s.root_stats()
    assert np.array_equal(rs["N"].sum(axis=1), np.full(G, playouts))
    for t in range(0, G, 7):
        _invariants(s.tree_dump(t), playouts)


@pytest.mark.gpu
@pytest.mark.parametrize("K,playouts", [(4, 50), (16, 200)])
def test_hip_search_width_k_exact_playouts_equals_oracle(rules_golden, K, playouts):
    """SearchEngine.search(width = k): exactly `playouts` simulations per tree, trees bit-identical to the oracle running
    the same budgeted schedule."""
    import torch
    from cchess_zero_amd.engine import SearchEngine
    g = rules_golden
    ok = [(g["boards"][i] == 1).any() and (g["boards"][i] == 8).any() and g["counts"][i] > 0 for i in range(len(g["boards"]))]
    idx = np.nonzero(ok)[0][::61][:32]
    G = len(idx)
    boards, side = g["boards"][idx], g["side"][idx]
    fk = fakenet.make_forward("pos", 9)

    def fwd_dev(planes):
        lg, v = fk(planes.float().cpu().numpy())
        return torch.from_numpy(lg).cuda(), torch.from_numpy(v).cuda()
    hip = SearchEngine(G, 60000, width=K)
    hip.reset(boards, side, np.zeros(G, np.int32))
    steps = hip.search(fwd_dev, playouts)
    orc = O.Search(G, 60000)
    orc.reset(boards, side, np.zeros(G, np.int32))
    osteps = _budget_search_oracle(orc, fk, K, playouts)
    st, nodes, sims, _ = (x.cpu().numpy() for x in hip.status())
    assert np.all(sims == playouts) and steps == osteps
    hs, os_ = hip.root_stats_host(), orc.root_stats()
    assert np.array_equal(hs["N"], os_["N"]) and np.array_equal(hs["Q"].view(np.uint32), os_["Q"].view(np.uint32))
    for t in range(0, G, 5):
        assert np.array_equal(hip.tree_dump(t), orc.tree_dump(t))


@pytest.mark.gpu
def test_search_width_16_from_a_fresh_root_reaches_its_budget():
    """Behind a FRESH root every descent of a k-wide step picks the same child (the reference never updates the root's N — quirk Q2:
    U = 0 at the root — and its virtual loss leaves Q alone, main.py:403-404), so all but one are abandoned and a lock-step
    completes ONE simulation, like the reference's sixteen coroutines queueing behind one expansion.  SearchEngine.search must
    still deliver exactly `playouts` (round 6: its cap of 4 n + 8 catch-up steps ended such a search at 38 of 96), with the real
    net, over two consecutive searches with a re-root in between — and so must MCTS_tree.main with search_threads = 16, which
    uses it."""
    import torch
    from cchess_zero_amd.engine import SearchEngine
    from cchess_zero_amd.net import PolicyValueNet
    net = PolicyValueNet(2, "cuda:0", torch.float16, seed=5, split="strict")
    b0 = O.fen_to_board(O.START_FEN)
    G, K, playouts = 3, 16, 96
    boards, side = np.tile(b0, (G, 1)), np.zeros(G, np.uint8)
    e = SearchEngine(G, 20000, plane_dtype=torch.float32, channels=14, width=K)
    e.reset(boards, side, None)
    for ply in range(2):
        steps = e.search(net.forward_device, playouts)
        a = e.root_stats_host()
        sims = e.status()[2].cpu().numpy()
        kept = 0 if ply == 0 else kept_visits            # a re-rooted tree keeps the visits of the subtree it moved into
        assert (sims == playouts).all() and (a["N"].sum(axis=1) == playouts + kept).all() and playouts // K <= steps <= playouts + 8
        for g in range(G):
            _invariants(e.tree_dump(g), playouts + int(np.atleast_1d(kept)[g if ply else 0]))
        n = a["N"].astype(np.int64).copy()
        n[np.arange(128)[None, :] >= a["count"].astype(np.int64)[:, None]] = -1
        best = n.argmax(axis=1)
        kept_visits = a["N"][np.arange(G), best] - 1       # the played child's visits, minus the one that expanded it
        e.advance(a["label"][np.arange(G), best].astype(np.uint16))
    import main as M
    tree = M.MCTS_tree(M.START_STATE, net.forward, 16)
    tree.main(tree._state, "w", 0, 80)
    assert sum(c.N for c in tree.root.child.values()) == 80
