"""Full-size runs (BASELINE.json configs[1]: 4096 games, playout 400, 7-block bf16 net) checked through
size-independent properties, plus an oracle cross-check on a subset fed the very same net outputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _bench_positions(G, seed):
    import bench
    from cchess_zero_amd.rules import Rules
    r = Rules()
    return bench.synth_positions(r, G, seed)


def _tree_invariants(rec):
    """rec: pre-order dump [depth, label, N, Wbits, Qbits, Pbits, nchildren].  For every expanded, visited node:
    N == 1 + sum(children N) (the visit that expanded it + one per descent through it); Q == W / N in float32."""
    n = len(rec)
    stack = []  # (index, remaining children, sum of child N)
    for i in range(n):
        d, lab, N, wb, qb, pb, nc = (int(x) for x in rec[i])
        if N > 0:
            W = np.int32(wb).view(np.float32)
            Q = np.int32(qb).view(np.float32)
            assert Q == np.float32(W / np.float32(N)) or (np.isnan(Q) and np.isnan(W))
            assert abs(float(Q)) <= 1.0 + 1e-6
        while stack and stack[-1][1] == 0:
            idx, _, s = stack.pop()
            pN, pnc = int(rec[idx][2]), int(rec[idx][6])
            if pnc > 0:
                assert pN == 1 + s, (idx, pN, s)
        if stack:
            stack[-1][1] -= 1
            stack[-1][2] += N
        if nc > 0:
            stack.append([i, nc, 0])
        elif nc == 0:
            pass
    while stack:
        idx, rem, s = stack.pop()
        if rem == 0 and int(rec[idx][6]) > 0:
            assert int(rec[idx][2]) == 1 + s


def test_config1_4096_games_playout400_properties():
    from cchess_zero_amd.engine import SearchEngine
    from cchess_zero_amd.net import PolicyValueNet
    from oracle import oracle as O
    G, playouts = 4096, 400
    boards, side, rr = _bench_positions(G, 7)
    net = PolicyValueNet(7, "cuda:0", torch.bfloat16, seed=0)
    eng = SearchEngine(G, (playouts + 2) * 80, plane_dtype=torch.bfloat16, channels=16)
    eng.reset(boards, side, rr)
    # oracle shadow for a subset, fed the same logits/values the GPU net produced
    sub = np.arange(0, G, 64)
    orc = O.Search(len(sub), (playouts + 2) * 80)
    orc.reset(boards.cpu().numpy()[sub], side.cpu().numpy()[sub], rr.cpu().numpy()[sub])
    for step in range(playouts + 1):
        mode = 0 if step == 0 else 1
        planes, need = eng.select(mode)
        logits, value = net.forward_device(planes)
        eng.expand_backup(logits, value)
        if step % 1 == 0:
            op, on = orc.select(mode)
            assert np.array_equal(on, need.cpu().numpy()[sub]), step
            if step % 50 == 0:   # planes agree too (bf16 x16 vs f32 x14 encodings of the same leaf)
                assert np.array_equal(planes[:, :, :, :14].float().cpu().numpy()[sub], op), step
            orc.expand_backup(logits.float().cpu().numpy()[sub], value.float().cpu().numpy()[sub])
    st, nodes, sims, depth = (x.cpu().numpy() for x in eng.status())
    assert not np.any(st), "no tree may be parked at this size"
    assert np.all(sims == playouts)
    rs = eng.root_stats_host()
    cnt = rs["count"].astype(int)
    # every simulation ends in exactly one root child's subtree
    assert np.array_equal(rs["N"].sum(axis=1), np.full(G, playouts))
    assert np.all(rs["N"] >= 0) and np.all(np.abs(rs["Q"]) <= 1.0 + 1e-6)
    # root children == legal moves of the root position, in order
    for g in range(0, G, 257):
        mv = O.legal_moves(boards[g].cpu().numpy(), int(side[g]))
        assert cnt[g] == len(mv) and np.array_equal(rs["label"][g, :len(mv)], mv)
    # oracle subset: identical visit counts and bit-identical W/Q/P
    os_ = orc.root_stats()
    for k in ("label", "N"):
        assert np.array_equal(rs[k][sub], os_[k])
    for k in ("Q", "W", "P"):
        assert np.array_equal(rs[k][sub].view(np.uint32), os_[k].view(np.uint32)), k
    # structural invariants of whole trees
    for g in (0, 1000, 4095):
        _tree_invariants(eng.tree_dump(g, 1 << 18))
    assert int(nodes.max()) < (playouts + 2) * 80


def test_search_is_deterministic():
    from cchess_zero_amd.engine import SearchEngine
    from cchess_zero_amd.net import PolicyValueNet
    G = 512
    boards, side, rr = _bench_positions(G, 3)
    net = PolicyValueNet(2, "cuda:0", torch.bfloat16, seed=0)
    out = []
    for rep in range(2):
        eng = SearchEngine(G, 8192, plane_dtype=torch.bfloat16, channels=16)
        eng.reset(boards, side, rr)
        eng.search(net.forward_device, 60)
        st = eng.root_stats()
        out.append({k: v.clone() for k, v in st.items()})
    for k in out[0]:
        assert torch.equal(out[0][k], out[1][k]), k


def test_rules_at_65536_positions_properties():
    """configs[3] scale for the rules kernels: list/mask consistency, incremental hash == from scratch, plane sums."""
    from cchess_zero_amd.rules import Rules
    r = Rules()
    G = 65536
    import bench
    boards, side, rr = bench.synth_positions(r, G, 11, max_ply=60)
    moves, count, mask = r.movegen(boards, side)
    cnt = count.to(torch.int64) & 0xFFFF
    assert int(cnt.max()) <= 128 and int(cnt.min()) >= 1
    # popcount(mask) == count and every listed label has its bit set
    m = mask.cpu().numpy().view(np.uint32)
    pop = np.unpackbits(m.view(np.uint8), axis=1).sum(axis=1)
    assert np.array_equal(pop, cnt.cpu().numpy())
    mv = moves.cpu().numpy().view(np.uint16)
    first = mv[:, 0].astype(np.int64)
    assert np.all((m[np.arange(G), first >> 5] >> (first & 31).astype(np.uint32)) & 1)
    # play the first legal move everywhere: incremental Zobrist == from scratch, side flips, piece count drops by captures
    h = r.hash(boards, side)
    b2, s2 = boards.clone(), side.clone()
    pieces0 = (b2 != 0).sum(dim=1)
    cap, term = r.apply_move(b2, s2, moves[:, 0].contiguous(), h)
    assert torch.equal(h, r.hash(b2, s2))
    assert torch.equal(s2, 1 - side)
    assert torch.equal(pieces0 - (cap != 0).to(pieces0.dtype), (b2 != 0).sum(dim=1))
    # planes: one-hot, and (quirk Q1) their sum counts the pieces on the first 82 cells plus the 8 duplicated cells
    p = r.encode_planes(boards, side, torch.bfloat16, 16).float()
    assert float(p.max()) == 1.0 and float(p[..., 14:].abs().sum()) == 0.0
    assert torch.equal(p.sum(dim=3).clamp(max=1), p.sum(dim=3))
