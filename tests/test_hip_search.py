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


@pytest.mark.parametrize("mode", ["pos", "signed"])
def test_search_vs_oracle_batch(rules_golden, mode):
    """256 trees from corpus positions, 3 plies x 48 playouts, device-resident loop; compared with the
    oracle: needs_eval + planes every step, root stats and whole-tree dumps after each ply."""
    from oracle import oracle as O
    g = rules_golden
    ok = [(g["boards"][i] == 1).any() and (g["boards"][i] == 8).any() and g["counts"][i] > 0 for i in range(len(g["boards"]))]
    idx = np.nonzero(ok)[0][::11][:256]
    G = len(idx)
    boards, side = g["boards"][idx], g["side"][idx]
    rr = (np.arange(G) * 7 % 61).astype(np.int32)
    rr[::5] = 57
    hip = _HipEngine(G, 20000)
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
