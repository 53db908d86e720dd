"""Batched self-play on top of the lock-step search engine (host side of K7 / rows a17-a18 of SURVEY §8).

Mirrors, for G games at once, cchess_main.get_action (main.py:1332-1358) and cchess_main.selfplay
(main.py:1493-1554) of the reference:
  * pi = softmax(log(N)/temperature) over the root children (temperature 1 -> N / sum N),
  * the played move is sampled from 0.75*pi + 0.25*Dirichlet(0.3) when exploring (main.py:1346),
  * every ply records (canonical state, pi scattered over the 2086 labels in canonical orientation —
    rank-flipped for black, main.py:1507-1512 — and the mover),
  * a game ends when a king is captured (z = +1 for the winner's plies, -1 for the loser's) or after
    60 plies without capture (z = 0), main.py:1532-1545.
Games are independent: with several GPUs every rank plays its own shard of games and only the
finished (s, pi, z) records are exchanged (parallel.py).

Records are kept packed (45-byte nibble board + mover + <=128 (label, prob) pairs + z); `to_dense`
expands them to the reference's training tuples (planes [9,10,14] f32, pi [2086] f32, z).
"""
import numpy as np
import torch

from ._lib import MAXMOVES, NLABELS, NSQ, tables

REC_MAXMOVES = MAXMOVES
# one fixed-size record (uint8 view): board 90 B, side 1 B, count 1 B, z int8 1 B, pad 1 B, labels 128 x u16, probs 128 x f16
REC_BYTES = 90 + 4 + 2 * REC_MAXMOVES + 2 * REC_MAXMOVES


def pack_records(boards, side, labels, probs, counts, z):
    """-> uint8 [n, REC_BYTES].  boards u8 [n,90]; labels u16 [n,128]; probs f32 [n,128]; counts; z in {-1,0,1}."""
    n = boards.shape[0]
    rec = np.zeros((n, REC_BYTES), np.uint8)
    rec[:, :90] = boards
    rec[:, 90] = side
    rec[:, 91] = counts
    rec[:, 92] = np.asarray(z, np.int8).view(np.uint8)
    rec[:, 94:94 + 256] = np.ascontiguousarray(labels, np.uint16).view(np.uint8).reshape(n, 256)
    rec[:, 94 + 256:] = np.ascontiguousarray(probs, np.float32).astype(np.float16).view(np.uint8).reshape(n, 256)
    return rec


def unpack_records(rec):
    rec = np.ascontiguousarray(rec, np.uint8).reshape(-1, REC_BYTES)
    n = rec.shape[0]
    return dict(boards=rec[:, :90].copy(), side=rec[:, 90].copy(), counts=rec[:, 91].copy(),
                z=rec[:, 92].copy().view(np.int8),
                labels=rec[:, 94:94 + 256].copy().view(np.uint16).reshape(n, 128),
                probs=rec[:, 94 + 256:].copy().view(np.float16).reshape(n, 128).astype(np.float32))


def canonical_planes(boards, side):
    """generate_inputs (main.py:531-557) on the host for record expansion: [n,9,10,14] f32 with quirk Q1."""
    boards = np.asarray(boards, np.uint8).reshape(-1, NSQ)
    n = boards.shape[0]
    b = boards.reshape(n, 10, 9)
    flip = np.asarray(side).astype(bool)
    fb = b[:, ::-1, :]
    fb = np.where(fb == 0, 0, np.where(fb > 7, fb - 7, fb + 7)).astype(np.uint8)
    canon = np.where(flip[:, None, None], fb, b).reshape(n, NSQ)
    planes = np.zeros((n, 9, 10, 14), np.float32)
    cells = (np.arange(9)[:, None] * 9 + np.arange(10)[None, :])  # the reference's 9-stride read (quirk Q1)
    code = canon[:, cells]  # [n,9,10]
    for c in range(14):
        planes[..., c] = (code == c + 1)
    return planes


def to_dense(rec):
    """Packed records -> (planes [n,9,10,14] f32, pi [n,2086] f32, z [n] f32): the tuples
    cchess_main.run extends data_buffer with (main.py:1234-1240)."""
    u = unpack_records(rec)
    n = len(u["side"])
    unflip = tables()["unflip"].astype(np.int64)
    pi = np.zeros((n, NLABELS), np.float32)
    for i in range(n):
        k = int(u["counts"][i])
        lab = u["labels"][i, :k].astype(np.int64)
        if u["side"][i]:
            lab = unflip[lab]  # rank-flipped labels for black, main.py:1507-1512
        pi[i, lab] = u["probs"][i, :k]
    return canonical_planes(u["boards"], u["side"]), pi, u["z"].astype(np.float32)


class SelfPlay:
    """G concurrent self-play games on one GPU."""

    def __init__(self, engine, net, playouts, exploration=True, temperature=1.0, seed=0, max_plies=512):
        self.eng, self.net = engine, net
        self.playouts = int(playouts)
        self.exploration = exploration
        self.temperature = float(temperature)
        self.max_plies = max_plies
        self.dev = engine.dev
        self.gen = torch.Generator(device=self.dev).manual_seed(seed)
        self.sims = 0

    def start(self, boards, side, rr=None):
        self.eng.reset(boards, side, rr)
        self.eng.compact = True   # finished games park their trees: their rows drop out of the net's batch
        G = self.eng.G
        self.active = torch.ones(G, dtype=torch.bool, device=self.dev)
        self.hist = [[] for _ in range(G)]   # per game: list of (board u8[90], side, labels u16[k], probs f32[k])
        self.result = np.zeros(G, np.int8)   # +1 red ('w') won, -1 black won, 0 draw / unfinished
        self.done = np.zeros(G, bool)
        self.plies = 0

    def policy_from_visits(self, st):
        """softmax(log(N)/T) over the root children (main.py:1339-1341); padding -> 0."""
        N = st["N"].to(torch.float64)
        cnt = (st["count"].to(torch.int64) & 0xFFFF).unsqueeze(1)
        valid = torch.arange(MAXMOVES, device=self.dev).unsqueeze(0) < cnt
        logit = torch.where(valid & (N > 0), torch.log(N.clamp(min=1)) / self.temperature, torch.full_like(N, -float("inf")))
        # all-zero visit rows (cannot happen after >=1 playout) fall back to uniform over the legal moves
        none = ~torch.isfinite(logit).any(dim=1, keepdim=True)
        logit = torch.where(none & valid, torch.zeros_like(logit), logit)
        pi = torch.softmax(logit, dim=1)
        return torch.where(valid, torch.nan_to_num(pi, nan=0.0), torch.zeros_like(pi)), valid

    def step_ply(self, forward=None):
        """One ply for every active game: search, sample, record, advance, adjudicate."""
        eng = self.eng
        fwd = forward or self.net.forward_device
        act = self.active.to(torch.uint8)
        eng.search(fwd, self.playouts, active=act)
        self.sims += int(self.active.sum().item()) * self.playouts
        st = eng.root_stats()
        pi, valid = self.policy_from_visits(st)
        p = pi
        if self.exploration:  # 0.75*pi + 0.25*Dirichlet(0.3 * ones(k)), main.py:1346
            g = torch._standard_gamma(torch.full(pi.shape, 0.3, dtype=torch.float64, device=self.dev), generator=self.gen)
            g = torch.where(valid, g, torch.zeros_like(g))
            d = g / g.sum(dim=1, keepdim=True).clamp(min=1e-300)
            p = 0.75 * pi + 0.25 * d
        p = torch.where(valid, torch.nan_to_num(p, nan=0.0, posinf=0.0, neginf=0.0), torch.zeros_like(p))
        # rows without any child (finished / parked games) still need a well-formed distribution for the
        # batched sampler; their draw is discarded below
        dead = p.sum(dim=1, keepdim=True) <= 0
        onehot0 = torch.zeros_like(p)
        onehot0[:, 0] = 1.0
        p = torch.where(dead, onehot0, p)
        p = p / p.sum(dim=1, keepdim=True)
        choice = torch.multinomial(p.clamp(min=0), 1, generator=self.gen).squeeze(1)
        played = st["label"].gather(1, choice.unsqueeze(1)).squeeze(1)
        played = torch.where(self.active & ~dead.squeeze(1), played, torch.full_like(played, -1))
        # host-side record of (state, pi, mover) for the active games
        rb, rs, _ = eng.root_state()
        rb, rs = rb.cpu().numpy(), rs.cpu().numpy()
        lab = st["label"].cpu().numpy().view(np.uint16)
        cnt = st["count"].cpu().numpy().view(np.uint16)
        pih = pi.float().cpu().numpy()
        act_h = self.active.cpu().numpy()
        for g_ in np.nonzero(act_h)[0]:
            k = int(cnt[g_])
            self.hist[g_].append((rb[g_].copy(), int(rs[g_]), lab[g_, :k].copy(), pih[g_, :k].copy()))
        eng.advance(played)
        self.plies += 1
        # adjudication on the new root positions, main.py:1532-1545
        nb, ns, nrr = eng.root_state()
        nb_h, nrr_h = nb.cpu().numpy(), nrr.cpu().numpy()
        K = (nb_h == 1).any(axis=1)
        k = (nb_h == 8).any(axis=1)
        for g_ in np.nonzero(act_h)[0]:
            if not K[g_] or not k[g_]:
                self.result[g_] = 1 if not k[g_] else -1
                self.done[g_] = True
            elif nrr_h[g_] >= 60 or len(self.hist[g_]) >= self.max_plies:
                self.result[g_] = 0
                self.done[g_] = True
        self.active = torch.from_numpy(~self.done).to(self.dev)
        return int(self.active.sum().item())

    def play(self, forward=None, max_plies=None):
        n = 0
        while bool(self.active.any().item()) and (max_plies is None or n < max_plies):
            self.step_ply(forward)
            n += 1
        return self.records()

    def records(self, only_finished=True):
        """Packed (s, pi, z) records of all (finished) games."""
        B, S, L, P, C, Z = [], [], [], [], [], []
        for g_, h in enumerate(self.hist):
            if only_finished and not self.done[g_]:
                continue
            for (b, s, lab, pr) in h:
                k = len(lab)
                l2 = np.full(128, 0xFFFF, np.uint16); l2[:k] = lab
                p2 = np.zeros(128, np.float32); p2[:k] = pr
                r = int(self.result[g_])
                z = 0 if r == 0 else (1 if (r == 1) == (s == 0) else -1)  # winner's plies +1, loser's -1
                B.append(b); S.append(s); L.append(l2); P.append(p2); C.append(k); Z.append(z)
        if not B:
            return np.zeros((0, REC_BYTES), np.uint8)
        return pack_records(np.stack(B), np.asarray(S, np.uint8), np.stack(L), np.stack(P), np.asarray(C, np.uint8), np.asarray(Z, np.int8))
