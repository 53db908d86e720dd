"""Batched, device-resident self-play on top of the lock-step search engine (rows a17-a18 of SURVEY §8).

Mirrors, for G games at once, cchess_main.get_action (main.py:1332-1358) and cchess_main.selfplay
(main.py:1493-1554) of the reference:
  * pi = softmax(log(N)/temperature) over the root children (temperature 1 -> N / sum N),
  * the played move is sampled from 0.75*pi + 0.25*Dirichlet(0.3) when exploring (main.py:1346),
  * every ply records (canonical state, pi scattered over the 2086 labels in canonical orientation —
    rank-flipped for black, main.py:1507-1512 — and the mover),
  * a game ends when a king is captured (z = +1 for the winner's plies, -1 for the loser's) or after
    60 plies without capture (z = 0), main.py:1532-1545.
All of it runs on the device (cz_selfplay_choose / cz_search_advance / cz_selfplay_adjudicate / cz_selfplay_flush,
csrc/cz_selfplay.hip): there is no per-ply host synchronisation and no per-game Python loop; finished games hand their
records to a device ring and — in continuous mode — their slot starts the next game at once, so the device stays full
for as long as the loop runs.  The host only drains the ring (drain()).  With several GPUs every rank plays its own
games and only the drained records are exchanged (parallel.py).

Records are packed (include/cchess_hip.h CZ_REC_*: board, mover, <=128 (label, visit count) pairs, z); `to_dense`
expands them to the reference's training tuples (planes [9,10,14] f32, pi [2086], z).  pi is recomputed from the
visit counts with the reference's own float64 expression, so it is bit-identical to what get_action returns.
"""
import ctypes as C

import numpy as np
import torch

from ._lib import (MAXMOVES, NLABELS, NSQ, REC_BYTES, REC_COUNT, REC_FLAGS, REC_LABELS, REC_PLY, REC_SIDE, REC_VISITS, REC_Z,
                   SP_STATS, check, lib, tables)

REC_MAXMOVES = MAXMOVES


def pack_records(boards, side, labels, visits, counts, z, ply=None):
    """-> uint8 [n, REC_BYTES].  boards u8 [n,90]; labels u16 [n,128]; visits [n,128] ints; counts; z in {-1,0,1}."""
    n = boards.shape[0]
    rec = np.zeros((n, REC_BYTES), np.uint8)
    rec[:, :NSQ] = boards
    rec[:, REC_SIDE] = side
    rec[:, REC_COUNT] = counts
    rec[:, REC_Z] = np.asarray(z, np.int8).view(np.uint8)
    v = np.asarray(visits, np.int64)
    rec[:, REC_FLAGS] = (v > 65535).any(axis=1)
    if ply is not None:
        rec[:, REC_PLY:REC_PLY + 2] = np.ascontiguousarray(ply, np.uint16).view(np.uint8).reshape(n, 2)
    rec[:, REC_LABELS:REC_LABELS + 256] = np.ascontiguousarray(labels, np.uint16).view(np.uint8).reshape(n, 256)
    rec[:, REC_VISITS:REC_VISITS + 256] = np.ascontiguousarray(np.minimum(v, 65535), np.uint16).view(np.uint8).reshape(n, 256)
    return rec


def unpack_records(rec):
    rec = np.ascontiguousarray(rec, np.uint8).reshape(-1, REC_BYTES)
    n = rec.shape[0]
    return dict(boards=rec[:, :NSQ].copy(), side=rec[:, REC_SIDE].copy(), counts=rec[:, REC_COUNT].copy(),
                z=rec[:, REC_Z].copy().view(np.int8), flags=rec[:, REC_FLAGS].copy(),
                ply=rec[:, REC_PLY:REC_PLY + 2].copy().view(np.uint16).reshape(n),
                labels=rec[:, REC_LABELS:REC_LABELS + 256].copy().view(np.uint16).reshape(n, 128),
                visits=rec[:, REC_VISITS:REC_VISITS + 256].copy().view(np.uint16).reshape(n, 128))


def canonical_boards(boards, side):
    """try_flip (main.py:560-574) on piece-code boards: for black reverse the rank order and swap the colours."""
    boards = np.asarray(boards, np.uint8).reshape(-1, NSQ)
    n = boards.shape[0]
    b = boards.reshape(n, 10, 9)
    flip = np.asarray(side).astype(bool)
    fb = b[:, ::-1, :]
    fb = np.where(fb == 0, 0, np.where(fb > 7, fb - 7, fb + 7)).astype(np.uint8)
    return np.where(flip[:, None, None], fb, b).reshape(n, NSQ)


def canonical_planes(boards, side):
    """generate_inputs (main.py:531-557) on the host for record expansion: [n,9,10,14] f32 with quirk Q1."""
    canon = canonical_boards(boards, side)
    n = canon.shape[0]
    planes = np.zeros((n, 9, 10, 14), np.float32)
    cells = (np.arange(9)[:, None] * 9 + np.arange(10)[None, :])  # the reference's 9-stride read (quirk Q1)
    code = canon[:, cells]  # [n,9,10]
    for c in range(14):
        planes[..., c] = (code == c + 1)
    return planes


def visit_policy(visits, temperature=1.0):
    """get_action's probs (main.py:1341, softmax :1111-1116) with the reference's own float64 expression:
    softmax(1.0 / temperature * np.log(visits)); zero visits -> log 0 = -inf -> probability 0."""
    with np.errstate(divide="ignore"):
        x = 1.0 / temperature * np.log(np.asarray(visits, dtype=np.int64))
    probs = np.exp(x - np.max(x))
    probs /= np.sum(probs)
    return probs


def to_dense(rec, temperature=1.0, exact=True):
    """Packed records -> (planes [n,9,10,14] f32, pi [n,2086] f64, z [n] f32): the tuples cchess_main.run extends
    data_buffer with (main.py:1234-1240).  exact: pi through visit_policy record by record (bit-identical to the
    reference); otherwise one vectorised pass (same values to the last ulp or two)."""
    u = unpack_records(rec)
    n = len(u["side"])
    unflip = tables()["unflip"].astype(np.int64)
    pi = np.zeros((n, NLABELS), np.float64)
    if exact:
        for i in range(n):
            k = int(u["counts"][i])
            if k == 0:
                continue
            lab = u["labels"][i, :k].astype(np.int64)
            if u["side"][i]:
                lab = unflip[lab]  # rank-flipped labels for black, main.py:1507-1512
            pi[i, lab] = visit_policy(u["visits"][i, :k], temperature)
    elif n:
        k = u["counts"].astype(np.int64)
        valid = np.arange(128)[None, :] < k[:, None]
        with np.errstate(divide="ignore"):
            x = np.where(valid, np.log(u["visits"].astype(np.float64)) / temperature, -np.inf)
        e = np.exp(x - x.max(axis=1, keepdims=True))
        p = e / e.sum(axis=1, keepdims=True)
        lab = u["labels"].astype(np.int64)
        lab = np.where(valid, lab, 0)
        lab = np.where(u["side"][:, None].astype(bool), unflip[lab], lab)
        rows = np.repeat(np.arange(n), 128).reshape(n, 128)
        pi[rows[valid], lab[valid]] = p[valid]
    return canonical_planes(u["boards"], u["side"]), pi, u["z"].astype(np.float32)


class SelfPlay:
    """G concurrent self-play games on one GPU, device-resident.

    continuous=True : a finished game's slot starts a new game at once (the production loop: the batch never decays);
    continuous=False: finished games are parked — play() returns when every game has ended (one game per slot).
    """

    def __init__(self, engine, net, playouts, exploration=True, temperature=1.0, seed=0, max_plies=512, ring_records=None,
                 continuous=True, eval_cache=False, xcache_log2=0):
        self.eng, self.net = engine, net
        self.playouts = int(playouts)
        self.exploration = bool(exploration)
        self.temperature = float(temperature)
        self.max_plies = int(max_plies)
        self.continuous = bool(continuous)
        # evaluation cache (engine.set_eval_cache): valid as long as the net's weights do not change — start() empties it
        self.eval_cache = bool(eval_cache)
        # its cross-tree level (engine.set_xcache): games from one start position share their openings
        self.xcache_log2 = int(xcache_log2) if eval_cache else 0
        self.dev = engine.dev
        self.gen = torch.Generator(device=self.dev).manual_seed(seed)
        self.ring_records = ring_records
        self.plies = 0
        self.lock_steps = 0
        if engine.width != 1:
            raise ValueError("SelfPlay drives one simulation in flight per tree (thousands of trees fill the net batch)")

    # -- setup -------------------------------------------------------------------------------------
    def start(self, boards, side, rr=None):
        """Fresh trees on the given positions; every later game of a slot starts from the same position."""
        eng = self.eng
        if self.eval_cache or eng.eval_cache:
            eng.set_eval_cache(self.eval_cache)   # turning it on empties it: new weights, new cache
            if self.eval_cache and (self.xcache_log2 or eng.xcache_log2):
                eng.set_xcache(self.xcache_log2)
        eng.reset(boards, side, rr)
        eng.compact = not self.continuous   # parked games drop out of the net's batch; a full batch needs no compaction
        G = eng.G
        check(lib().cz_selfplay_begin(eng.ctx.h, self.max_plies, None, None, None), "cz_selfplay_begin")
        p = C.c_void_p()
        check(lib().cz_selfplay_active(eng.ctx.h, C.byref(p)), "cz_selfplay_active")
        self._active_ptr = p
        R = int(self.ring_records or max(65536, 64 * G))
        self.ring = torch.zeros((R, REC_BYTES), dtype=torch.uint8, device=self.dev)
        self.cursor = torch.zeros(1, dtype=torch.int64, device=self.dev)       # records ever handed to the ring
        self.read_cursor = torch.zeros(1, dtype=torch.int64, device=self.dev)  # records the host has drained
        self._read = 0
        self.played = torch.empty(G, dtype=torch.int16, device=self.dev)
        self.fin_n = torch.zeros(G, dtype=torch.int32, device=self.dev)
        self._alpha = torch.full((G, MAXMOVES), 0.3, dtype=torch.float32, device=self.dev)
        self._stats = torch.zeros(len(SP_STATS), dtype=torch.int64, device=self.dev)
        self.plies = 0
        self.lock_steps = 0
        self._dropped_seen = 0
        self.overflow_intervals = 0

    # -- one ply of every game ------------------------------------------------------------------------
    def step_ply(self, forward=None, forced=None):
        """Search, choose, record, advance, adjudicate, flush — all enqueued, nothing synchronised.
        forced: int16/uint16 [G] labels overriding the sampled moves (0xFFFF = sample), for replaying recorded games."""
        eng = self.eng
        fwd = forward or self.net.forward_device
        eng.search(fwd, self.playouts, active=None if self.continuous else self._active_ptr)
        self._transition(0, forced)
        self.plies += 1

    def _transition(self, min_sims, forced=None):
        """choose / advance / adjudicate / flush for every game (min_sims = 0) or for the games whose search has completed
        min_sims simulations (asynchronous plies)."""
        eng, h = self.eng, self.eng.ctx.h
        gamma = None
        if self.exploration:   # np.random.dirichlet(0.3 * ones(k)) = normalised Gamma(0.3) variates, main.py:1346
            gamma = torch._standard_gamma(self._alpha, generator=self.gen)
        u = torch.rand(eng.G, generator=self.gen, device=self.dev, dtype=torch.float32)
        f = None
        if forced is not None:
            f = torch.as_tensor(np.ascontiguousarray(forced).view(np.int16) if not torch.is_tensor(forced) else forced).to(self.dev).contiguous()
        vp = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        eng.ctx.bind_stream()
        check(lib().cz_selfplay_choose(h, vp(gamma), vp(u), vp(f), self.temperature, 0.25 if self.exploration else 0.0,
                                       int(min_sims), vp(self.played)), "cz_selfplay_choose")
        eng.advance(self.played)
        check(lib().cz_selfplay_adjudicate(h, 1 if self.continuous else 0, vp(self.played) if min_sims > 0 else None,
                                           vp(self.fin_n)), "cz_selfplay_adjudicate")
        csum = torch.cumsum(self.fin_n, 0, dtype=torch.int64)
        offset = csum - self.fin_n + self.cursor
        check(lib().cz_selfplay_flush(h, vp(self.fin_n), vp(offset), vp(self.ring), self.ring.shape[0], vp(self.read_cursor)),
              "cz_selfplay_flush")
        self.cursor += csum[-1:]

    def run(self, plies, forward=None):
        for _ in range(int(plies)):
            self.step_ply(forward)

    def run_async(self, steps, forward=None, every=8, terminal_extra=4):
        """ASYNCHRONOUS plies (continuous mode): `steps` lock-steps of the net, every game at its own pace.  Each step is
        one select / net / expand for all slots; every `every` steps the games whose search has completed its `playouts`
        simulations choose, move, are adjudicated and — if finished — re-seeded.  With terminal_extra > 0 simulations
        that end on terminal / drawn leaves complete inside the select launch without a net row, so a game finishes its
        search in fewer steps than it has playouts and the net batch is never spent on leaves that need no evaluation.
        Per tree the simulations are the same sequence as in lock-step plies (trees bit-identical for the same evaluations)."""
        assert self.continuous, "asynchronous plies re-seed finished slots at once"
        eng = self.eng
        fwd = forward or self.net.forward_device
        prev_extra = eng.terminal_extra
        eng.set_terminal_extra(terminal_extra)
        eng.set_sim_target(self.playouts)
        try:
            for i in range(int(steps)):
                eng.step(fwd, mode=1)            # an unexpanded root (fresh game, new ply) is expanded by this step
                self.lock_steps += 1
                if (i + 1) % every == 0:
                    self._transition(self.playouts)
        finally:
            eng.set_sim_target(0)
            eng.set_terminal_extra(prev_extra)

    def play(self, forward=None, max_plies=None):
        """continuous=False: play until every game has ended (or max_plies plies); returns the drained records.
        continuous=True needs max_plies (the loop never runs out of games)."""
        if self.continuous and max_plies is None:
            raise ValueError("a continuous self-play loop has no natural end: pass max_plies (or drive step_ply / drain yourself)")
        n = 0
        out = []
        while max_plies is None or n < max_plies:
            self.step_ply(forward)
            n += 1
            if n % 8 == 0 or (max_plies is not None and n >= max_plies):
                out.append(self.drain())
                if not self.continuous and not bool(self.active().any().item()):
                    break
        out.append(self.drain())
        return np.concatenate(out, axis=0) if out else np.zeros((0, REC_BYTES), np.uint8)

    # -- host side --------------------------------------------------------------------------------------
    def active(self):
        """uint8 [G] host tensor: 1 = game in progress (synchronises)."""
        self.eng.ctx.bind_stream()
        buf = np.empty(self.eng.G, np.uint8)
        check(lib().cz_download(self.eng.ctx.h, buf.ctypes.data_as(C.c_void_p), self._active_ptr, self.eng.G), "cz_download")
        return torch.from_numpy(buf)

    def drain_device(self, on_overflow="raise"):
        """-> uint8 [n, REC_BYTES] DEVICE tensor with the records finished since the last drain (synchronises on the
        cursor and the drop counter).  The returned rows stay valid until the ring wraps over them.
        If finished games have been DROPPED since the last drain because the ring was full of undrained rows
        (cz_selfplay_flush skips such a game but the cursor still advances by its length, so the interval contains slots
        that were never written), the whole interval is discarded — never handed out — and, with on_overflow = "raise"
        (default), a RuntimeError says so; on_overflow = "skip" counts it in self.overflow_intervals and returns no rows
        (a long training run prefers losing one interval of samples to stopping).  Drain more often or pass a larger
        ring_records to avoid it."""
        self.eng.ctx.bind_stream()
        check(lib().cz_selfplay_stats(self.eng.ctx.h, C.c_void_p(self._stats.data_ptr())), "cz_selfplay_stats")
        dropped = int(self._stats[SP_STATS.index("dropped")].item())
        c = int(self.cursor.item())
        if dropped > self._dropped_seen:
            n_new = dropped - self._dropped_seen
            self._dropped_seen = dropped
            self._read = c
            self.read_cursor.fill_(c)
            self.overflow_intervals += 1
            msg = ("self-play record ring overflow: %d records of finished games were dropped since the last drain "
                   "(ring of %d records; drain more often or raise ring_records); the interval's records are discarded"
                   % (n_new, self.ring.shape[0]))
            if on_overflow == "raise":
                raise RuntimeError(msg)
            print("WARNING:", msg)
            return self.ring[:0]
        r, R = self._read, self.ring.shape[0]
        n = min(c - r, R)
        if n <= 0:
            return self.ring[:0]
        lo = (c - n) % R
        out = self.ring[lo:lo + n] if lo + n <= R else torch.cat([self.ring[lo:], self.ring[:lo + n - R]], 0)
        self._read = c
        self.read_cursor.fill_(c)
        return out

    def drain(self):
        """-> uint8 [n, REC_BYTES] host array of the records finished since the last drain."""
        return self.drain_device().cpu().numpy().reshape(-1, REC_BYTES)

    def stats(self):
        """Running totals since start(): games, red_wins, black_wins, draws, plies (records), stalled, dropped, sims."""
        self.eng.ctx.bind_stream()
        check(lib().cz_selfplay_stats(self.eng.ctx.h, C.c_void_p(self._stats.data_ptr())), "cz_selfplay_stats")
        s = self._stats.cpu().numpy()
        d = {k: int(v) for k, v in zip(SP_STATS, s)}
        d["sims"] += int(self.eng.status()[2].sum().item())   # + the simulations of the searches in progress
        d["plies_played"] = self.plies          # lock-step plies (step_ply)
        d["lock_steps"] = self.lock_steps       # select / net / expand steps of the asynchronous loop (run_async)
        return d
