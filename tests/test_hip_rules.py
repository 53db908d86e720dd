"""GPU parity of the rules kernels K1-K3 against the golden vectors of the unmodified reference
and against the C oracle on a larger seeded corpus.  Bit-exact (integer / byte work)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rules():
    from cchess_zero_amd.rules import Rules
    return Rules()


def _u16(t):
    return t.cpu().numpy().view(np.uint16)


def test_movegen_golden(rules, rules_golden):
    """Ordered move lists, counts and 2086-bit masks of 4 381 reference positions (k_movegen_list<MASK>: one position per lane,
    czm_list); and the same masks and counts from the mask-only kernel (k_movegen_mask: one position per lane, no list)."""
    g = rules_golden
    moves, count, mask = rules.movegen(g["boards"], g["side"])
    moves, count, mask = _u16(moves), _u16(count), mask.cpu().numpy().view(np.uint32)
    assert np.array_equal(count, g["counts"])
    assert np.array_equal(moves, g["moves"])  # ordered list incl. 0xFFFF padding
    # mask == set of the listed labels
    exp = np.zeros_like(mask)
    for i in range(len(count)):
        for l in g["moves"][i, :count[i]]:
            exp[i, l >> 5] |= np.uint32(1) << np.uint32(l & 31)
    assert np.array_equal(mask, exp)
    none, count2, mask2 = rules.movegen(g["boards"], g["side"], want_moves=False)
    assert none is None and np.array_equal(_u16(count2), g["counts"])
    assert np.array_equal(mask2.cpu().numpy().view(np.uint32), exp)
    _, count3, nomask = rules.movegen(g["boards"], g["side"], want_mask=False, want_moves=False)   # counts only
    assert nomask is None and np.array_equal(_u16(count3), g["counts"])


def test_movegen_open_boards_with_up_to_100_moves(rules):
    """Sparse synthetic boards (tests/conftest.py::open_boards, 33-100+ moves per position: twice what playouts from the start
    position reach) through all four instantiations of k_movegen_list and through k_movegen_mask, against the C oracle: the long
    end of the 128-slot rows, where a list slot is addressed from the ignore slot at 128."""
    from conftest import open_boards
    from oracle import oracle as O
    boards, side = open_boards(900, 12)
    want = [np.asarray(O.legal_moves(boards[i], int(side[i])), np.uint16) for i in range(len(boards))]
    n = np.array([len(w) for w in want])
    assert n.max() >= 95 and n.mean() > 55
    exp_mask = np.zeros((len(boards), 66), np.uint32)
    for i, w in enumerate(want):
        np.bitwise_or.at(exp_mask[i], w.astype(np.int64) >> 5, np.uint32(1) << (w.astype(np.uint32) & np.uint32(31)))
    for want_mask in (True, False):
        for pad in (True, False):
            mv, cnt, m = rules.movegen(boards, side, want_mask=want_mask, pad=pad)
            mv, cnt = _u16(mv), _u16(cnt)
            assert np.array_equal(cnt, n)
            for i, w in enumerate(want):
                assert np.array_equal(mv[i, :len(w)], w), (i, pad, want_mask)
                if pad:
                    assert (mv[i, len(w):] == 0xFFFF).all()
            if want_mask:
                assert np.array_equal(m.cpu().numpy().view(np.uint32), exp_mask)
    _, cnt, m = rules.movegen(boards, side, want_moves=False)
    assert np.array_equal(_u16(cnt), n) and np.array_equal(m.cpu().numpy().view(np.uint32), exp_mask)


def test_movegen_golden_no_pad(rules, rules_golden):
    """CZ_MOVES_NO_PAD (cz_movegen_ex): the same 4 381 reference lists, counts and masks with rows written
    up to their count only.  The buffer is pre-filled with a sentinel: a row holds the golden labels up to its count, the
    sentinel from the next 16-byte piece on (nothing is written there), and never 0xFFFF padding beyond the last piece."""
    import ctypes as C
    from cchess_zero_amd._lib import check, lib
    g = rules_golden
    G = len(g["counts"])
    for want_mask in (True, False):
        boards = torch.from_numpy(g["boards"]).cuda().contiguous()
        side = torch.from_numpy(g["side"]).cuda().contiguous()
        moves = torch.full((G, 128), 0x5A5A, dtype=torch.int16, device="cuda")
        count = torch.empty(G, dtype=torch.int16, device="cuda")
        mask = torch.empty((G, 66), dtype=torch.int32, device="cuda") if want_mask else None
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        rules.ctx.bind_stream()
        check(lib().cz_movegen_ex(rules.ctx.h, p(boards), p(side), G, p(moves), p(count), p(mask), 1), "cz_movegen_ex")
        mv, cnt = _u16(moves), _u16(count)
        assert np.array_equal(cnt, g["counts"])
        col = np.arange(128)[None, :]
        valid = col < cnt[:, None]
        assert np.array_equal(mv[valid], g["moves"][valid])
        untouched = col >= ((cnt[:, None] + 7) // 8) * 8          # behind the last 16-byte piece of a row
        assert (mv[untouched] == 0x5A5A).all() and untouched.sum() > 0.6 * mv.size
        if want_mask:
            _, _, m0 = rules.movegen(g["boards"], g["side"], want_moves=False)
            assert torch.equal(mask, m0)
    # the Python wrapper: pad=False is the same call; strict=True raises on a board that is not a Xiangqi set
    mv2, cnt2, _ = rules.movegen(g["boards"][:100], g["side"][:100], want_mask=False, pad=False)
    assert np.array_equal(_u16(cnt2), g["counts"][:100])
    for i in range(100):
        assert np.array_equal(_u16(mv2)[i, :g["counts"][i]], g["moves"][i, :g["counts"][i]])
    bad = g["boards"][:4].copy()
    bad[2, :] = 0
    bad[2, :3] = 3 if g["side"][2] == 0 else 10               # three rooks of the side to move
    from cchess_zero_amd._lib import CchessHipError
    with pytest.raises(CchessHipError, match="not a Xiangqi set"):
        rules.movegen(bad, g["side"][:4], strict=True)


def test_apply_move_golden(rules, rules_golden):
    g = rules_golden
    boards = torch.from_numpy(g["boards"].copy()).cuda()
    side = torch.from_numpy(g["side"].copy()).cuda()
    h = rules.hash(boards, side)
    cap, term = rules.apply_move(boards, side, g["chosen"].view(np.int16), h)
    nb = boards.cpu().numpy()
    has = g["chosen"] != 0xFFFF
    assert np.array_equal(nb[has], g["next_boards"][has])
    assert np.array_equal(nb[~has], g["boards"][~has])
    assert np.array_equal((cap.cpu().numpy() != 0)[has].astype(np.int8), g["kill"][has])
    assert np.array_equal(side.cpu().numpy()[has], 1 - g["side"][has])
    t = term.cpu().numpy()
    K = (nb == 1).any(axis=1)
    k = (nb == 8).any(axis=1)
    assert np.array_equal(t & 1, (~K).astype(np.int8)) and np.array_equal((t >> 1) & 1, (~k).astype(np.int8))
    # incremental Zobrist == from scratch on the new position
    h2 = rules.hash(boards, side)
    assert torch.equal(h[torch.from_numpy(has).cuda()], h2[torch.from_numpy(has).cuda()])


def test_planes_golden(rules, rules_golden):
    g = rules_golden
    p32 = rules.encode_planes(g["boards"], g["side"], torch.float32, 14).cpu().numpy()
    bits = np.packbits(p32.reshape(len(p32), -1) > 0.5, axis=1)
    assert np.array_equal(bits, g["planes_bits"])
    assert set(np.unique(p32)) <= {0.0, 1.0}
    p16 = rules.encode_planes(g["boards"], g["side"], torch.bfloat16, 16)
    assert p16.shape == (len(p32), 9, 10, 16)
    assert torch.equal(p16[..., :14].float().cpu(), torch.from_numpy(p32))
    assert float(p16[..., 14:].abs().sum()) == 0.0
    ph = rules.encode_planes(g["boards"], g["side"], torch.float16, 16)   # CZ_F16: 1.0 = 0x3C00
    assert torch.equal(ph[..., :14].float().cpu(), torch.from_numpy(p32)) and float(ph[..., 14:].abs().sum()) == 0.0


def test_rules_vs_oracle_large_corpus(rules):
    """Seeded random playouts driven entirely on the GPU (movegen -> pick -> apply), every position
    cross-checked against the C oracle: ordered moves, next board, hash, planes; and at every ply the mask-only kernel's
    masks and counts for ALL 2 048 positions against the list kernel's (82 k positions)."""
    from oracle import oracle as O
    rng = np.random.default_rng(1234)
    G = 2048
    boards = torch.from_numpy(np.tile(O.fen_to_board(O.START_FEN), (G, 1))).cuda()
    side = torch.zeros(G, dtype=torch.uint8).cuda()
    zt, zs = O.zobrist_table()
    checked = 0
    for ply in range(40):
        moves, count, lmask = rules.movegen(boards, side)
        _, mcount, mmask = rules.movegen(boards, side, want_moves=False)
        assert torch.equal(mcount, count) and torch.equal(mmask, lmask)
        hb, hs = boards.cpu().numpy(), side.cpu().numpy()
        mv, cnt = _u16(moves), _u16(count)
        planes = rules.encode_planes(boards, side).cpu().numpy()
        hh = rules.hash(boards, side).cpu().numpy().view(np.uint64)
        pick = np.full(G, 0xFFFF, np.uint16)
        sample = rng.choice(G, 96, replace=False) if ply % 3 else np.arange(G)
        for gi in sample:
            om = O.legal_moves(hb[gi], int(hs[gi]))
            assert cnt[gi] == len(om) and np.array_equal(mv[gi, :len(om)], om)
            assert np.array_equal(planes[gi], O.encode_planes(hb[gi], int(hs[gi])))
            assert int(hh[gi]) == O.zhash(hb[gi], int(hs[gi]))
            checked += 1
        for gi in range(G):
            if cnt[gi] and (hb[gi] == 1).any() and (hb[gi] == 8).any():
                pick[gi] = mv[gi, rng.integers(cnt[gi])]
        rules.apply_move(boards, side, pick.view(np.int16))
        nb = boards.cpu().numpy()
        for gi in sample[:64]:
            if pick[gi] != 0xFFFF:
                assert np.array_equal(nb[gi], O.apply_move(hb[gi], int(pick[gi]))[0])
    assert checked > 5000


def test_mask_kernel_persistent_waves_walk_many_groups(rules, rules_golden):
    """k_movegen_mask launches at most 2048 waves; a batch of more than 131 072 positions makes every wave walk several groups
    with the next group's boards prefetched into registers.  300 011 positions (4 688 groups, a ragged last one of 11) cut from
    the golden corpus at a stride that is no multiple of 64, so a position's lane and group change with its index: masks and
    counts equal the list kernel's for every position, and the goldens' counts where a position is a golden one."""
    b0 = np.asarray(rules_golden["boards"], np.uint8)
    s0 = np.asarray(rules_golden["side"], np.uint8)
    n0 = len(b0)
    G = 300011
    idx = (np.arange(G, dtype=np.int64) * 37) % n0
    boards = torch.from_numpy(b0[idx]).cuda()
    side = torch.from_numpy(s0[idx]).cuda()
    _, mcount, mmask = rules.movegen(boards, side, want_moves=False)
    ref_c = torch.empty(G, dtype=mcount.dtype, device="cuda")
    ref_m = torch.empty_like(mmask)
    for a in range(0, G, 65536):   # the list kernel (k_movegen_list<MASK>) in slices of 1 024 groups of 64 positions
        _, c, m = rules.movegen(boards[a:a + 65536], side[a:a + 65536])
        ref_c[a:a + 65536] = c
        ref_m[a:a + 65536] = m
    assert torch.equal(mcount, ref_c)
    assert torch.equal(mmask, ref_m)
    gc = np.asarray(rules_golden["counts"]).astype(np.int64)[idx]
    assert np.array_equal(_u16(mcount).astype(np.int64), gc)


def test_rules_edge_sizes(rules):
    """Empty batch and a single position."""
    from oracle import oracle as O
    m, c, k = rules.movegen(np.zeros((0, 90), np.uint8), np.zeros(0, np.uint8))
    assert m.shape == (0, 128) and c.shape == (0,)
    b = O.fen_to_board(O.START_FEN)[None]
    m, c, k = rules.movegen(b, np.zeros(1, np.uint8))
    assert int(_u16(c)[0]) == 44


@pytest.mark.parametrize("G", [1, 3, 4, 5, 63, 64, 65, 130, 4097])
def test_movegen_ragged_sizes_and_alignment_raw_abi(rules, rules_golden, G):
    """cz_movegen through the raw C-ABI on batch sizes around the kernels' group sizes (4 positions per wave for the list
    kernel, 64 for the mask-only kernel), with the boards at a 16-byte-aligned, an even and an ODD byte address (the ABI promises
    byte alignment only), and the mask at a 16-byte-aligned and a 4-byte-aligned address: list + mask, list only, mask only,
    count only — against the golden lists; rows beyond the batch are not touched."""
    from cchess_zero_amd._lib import check, lib
    from cchess_zero_amd.engine import _ptr
    g = rules_golden
    idx = (np.arange(G) * 37) % len(g["boards"])
    exp = np.zeros((G, 66), np.uint32)
    for i in range(G):
        for l in g["moves"][idx[i], :g["counts"][idx[i]]]:
            exp[i, l >> 5] |= np.uint32(1) << np.uint32(l & 31)
    for off in (0, 2, 1):
        raw = torch.zeros(G * 90 + 16, dtype=torch.uint8, device="cuda")
        raw[off:off + G * 90] = torch.from_numpy(g["boards"][idx].reshape(-1)).cuda()
        boards = raw[off:off + G * 90]
        side = torch.from_numpy(g["side"][idx]).cuda()
        for moff in (0, 1):
            for want_moves, want_mask in ((True, True), (True, False), (False, True), (False, False)):
                moves = torch.full((G + 2, 128), 0x1234, dtype=torch.int16, device="cuda")
                count = torch.full((G + 2,), 0x1234, dtype=torch.int16, device="cuda")
                mraw = torch.full(((G + 2) * 66 + 4,), 0x55, dtype=torch.int32, device="cuda")
                mask = mraw[moff:moff + (G + 2) * 66].view(G + 2, 66)
                check(lib().cz_movegen(rules.ctx.h, _ptr(boards), _ptr(side), G, _ptr(moves) if want_moves else None, _ptr(count),
                                       _ptr(mask) if want_mask else None), "cz_movegen")
                mv, ct = _u16(moves), _u16(count)
                assert np.array_equal(ct[:G], g["counts"][idx]) and (ct[G:] == 0x1234).all()
                if want_moves:
                    assert np.array_equal(mv[:G], g["moves"][idx]) and (mv[G:] == 0x1234).all()
                else:
                    assert (mv == 0x1234).all()
                mk = mask.cpu().numpy().view(np.uint32)
                if want_mask:
                    assert np.array_equal(mk[:G], exp) and (mk[G:] == 0x55).all()
                else:
                    assert (mk == 0x55).all()


@pytest.mark.parametrize("G", [1, 63, 64, 65, 4097, 150001])
def test_hash_ragged_sizes_and_alignment_raw_abi(rules, rules_golden, G):
    """cz_hash (one lane = one position, 64 positions staged per wave, persistent waves from 147 456 positions on) through the
    raw C-ABI on batch sizes around its group size, with the boards at a 16-byte-aligned, an even and an ODD byte address:
    every key against the C oracle's (sampled for the large batch); keys beyond the batch are not touched."""
    from oracle import oracle as O
    from cchess_zero_amd._lib import check, lib
    from cchess_zero_amd.engine import _ptr
    g = rules_golden
    idx = (np.arange(G) * 37) % len(g["boards"])
    hb, hs = g["boards"][idx], g["side"][idx]
    sample = np.arange(G) if G <= 4097 else np.unique(np.concatenate([np.arange(200), np.arange(G - 200, G), (np.arange(800) * 187) % G]))
    exp = np.array([O.zhash(hb[i], int(hs[i])) for i in sample], dtype=np.uint64)
    for off in (0, 2, 1):
        raw = torch.zeros(G * 90 + 16, dtype=torch.uint8, device="cuda")
        raw[off:off + G * 90] = torch.from_numpy(hb.reshape(-1)).cuda()
        boards = raw[off:off + G * 90]
        side = torch.from_numpy(hs).cuda()
        out = torch.full((G + 2,), 0x1234, dtype=torch.int64, device="cuda")
        check(lib().cz_hash(rules.ctx.h, _ptr(boards), _ptr(side), G, _ptr(out)), "cz_hash")
        h = out.cpu().numpy().view(np.uint64)
        assert np.array_equal(h[sample], exp)
        assert (h[G:] == 0x1234).all()
        if off == 0:
            first = h[:G].copy()
        else:
            assert np.array_equal(h[:G], first)   # every key, not only the sampled ones, is independent of the address


def test_movegen_mask_kernel_flags_unexpressible_positions(rules):
    """Both kernels answer 0xFFFF for a position the move vocabulary cannot express: 17 pieces of the side to move; an advisor
    off the palace's diagonal points whose step (e2 -> d1) has no label."""
    from oracle import oracle as O
    b = O.fen_to_board(O.START_FEN)
    many = b.copy()
    many[4 * 9 + 4] = 6               # a sixth red pawn on e4: 17 red pieces
    adv = b.copy()
    adv[3], adv[2 * 9 + 4] = 0, 2     # the red advisor from d0 to e2: e2 -> d1 / f1 are inside the palace and unlabelled
    boards = np.stack([b, many, adv])
    side = np.zeros(3, np.uint8)
    _, c_list, _ = rules.movegen(boards, side)
    _, c_mask, _ = rules.movegen(boards, side, want_moves=False)
    cl, cm = _u16(c_list), _u16(c_mask)
    assert cl[0] == 44 and cm[0] == 44
    assert cl[1] == 0xFFFF and cm[1] == 0xFFFF
    assert cl[2] == 0xFFFF and cm[2] == 0xFFFF


def test_non_xiangqi_boards_answer_0xffff(rules):
    """include/cchess_hip.h (cz_movegen): a board with more of a kind than a Xiangqi set holds for the side to move answers count
    0xFFFF from both stand-alone kernels (the one-lane-per-position generators would otherwise drop that piece's moves
    silently); with the other side to move the same board is generated normally, and list and set agree."""
    from oracle import oracle as O
    start = O.fen_to_board(O.START_FEN)
    boards, sides, bad = [], [], []
    for code in (3, 7, 5, 2, 4, 6, 1):
        b = start.copy()
        b[4 * 9 + 4] = code
        for s in (0, 1):
            boards.append(b.copy()); sides.append(s); bad.append(s == 0)
    bt, st = torch.from_numpy(np.stack(boards)).cuda(), torch.tensor(sides, dtype=torch.uint8).cuda()
    mv, c, m = rules.movegen(bt, st)
    _, c2, m2 = rules.movegen(bt, st, want_moves=False)
    c, c2 = _u16(c), _u16(c2)
    for i, is_bad in enumerate(bad):
        assert (c[i] == 0xFFFF) == is_bad and (c2[i] == 0xFFFF) == is_bad, (i, c[i], c2[i])
        if not is_bad:
            want = O.legal_moves(boards[i], sides[i])
            assert c[i] == len(want) == c2[i] and np.array_equal(_u16(mv)[i, :c[i]], np.asarray(want, np.uint16))
            assert torch.equal(m[i], m2[i])
