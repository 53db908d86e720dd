"""Lock-step search engine over G game trees (host side of K1-K7).

PyTorch is used only for device memory / streams; every search operation is one HIP kernel
launched through the C-ABI (include/cchess_hip.h).  Mirrors, for G trees at once, what the
reference's MCTS_tree (main.py:234-577) does for one.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import BF16, F16, F32, MAXMOVES, NLABELS, NSQ, check, lib


def _ptr(t):
    if t is None or isinstance(t, C.c_void_p):   # a raw device pointer (e.g. cz_selfplay_active) passes through
        return t
    return C.c_void_p(t.data_ptr())


def _mask(active, dev):
    """active mask argument -> something _ptr accepts: None, a raw device pointer, or a uint8 device tensor."""
    if active is None or isinstance(active, C.c_void_p):
        return active
    return torch.as_tensor(active).to(dev).to(torch.uint8).contiguous()


class Context:
    """Owns a cz_ctx bound to one GPU and to torch's current stream on it."""

    def __init__(self, max_games, max_nodes_per_tree, device=0):
        if not torch.cuda.is_available():
            raise _lib.CchessHipError("no HIP device visible: the cchess_hip path needs an MI355X (no CPU fallback)")
        self.device = torch.device("cuda", device)
        self.max_games = int(max_games)
        self.cap = int(max_nodes_per_tree)
        h = C.c_void_p()
        check(lib().cz_create(device, self.max_games, self.cap, C.byref(h)), "cz_create")
        self.h = h
        self.bind_stream()

    def bind_stream(self, stream=None):
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        check(lib().cz_set_stream(self.h, C.c_void_p(s.cuda_stream)), "cz_set_stream")

    def synchronize(self):
        check(lib().cz_synchronize(self.h), "cz_synchronize")

    def close(self):
        if getattr(self, "h", None):
            lib().cz_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SearchEngine:
    """G trees searched in lock-step: select -> (batched net) -> expand_backup."""

    def __init__(self, max_games, max_nodes_per_tree, device=0, plane_dtype=torch.float32, channels=14, ctx=None, width=1):
        """width = simulations in flight per tree and step (the reference's search_threads); 1 = sequential."""
        self.ctx = ctx or Context(max_games, max_nodes_per_tree, device)
        self.width = int(width)
        if self.width > 1:
            check(lib().cz_search_set_width(self.ctx.h, self.width), "cz_search_set_width")
        self.dev = self.ctx.device
        self.G = 0
        self.channels = int(channels)
        self.plane_dtype = plane_dtype
        self._pd = {torch.bfloat16: BF16, torch.float16: F16}.get(plane_dtype, F32)
        # step(): compact evaluation batches on the fused-net path.  Off by default: at 8192 trees the trunk needs
        # ceil(rows / 1024) workgroup rounds, so dropping the ~9 % terminal leaves of the bench workload removes no round
        # (and the row counter costs 60 us of atomics); self-play switches it on, where finished games park their trees.
        self.compact = False
        self.planes = None
        self.need = None
        self.terminal_extra = 0
        self.eval_cache = False
        self.xcache_log2 = 0

    def set_terminal_extra(self, n):
        """Terminal simulations (king captured / 60-ply rule: no net evaluation needed, main.py:409-416) a tree may complete
        inside one select launch before it presents a leaf that does need the net (cz_search_set_terminal_extra).  Trees
        stay bit-identical; a lock-step then completes more than one simulation per net row.  Width 1 only."""
        assert self.width == 1 or int(n) == 0, "terminal_extra applies to the one-simulation-per-tree select"
        check(lib().cz_search_set_terminal_extra(self.ctx.h, int(n)), "cz_search_set_terminal_extra")
        self.terminal_extra = int(n)

    def set_eval_cache(self, on):
        """Evaluation cache (include/cchess_hip.h: cz_search_set_eval_cache): a leaf whose position the tree has
        evaluated before is expanded from the remembered node inside the select launch, without a net row.  A hit is
        taken only when the stored position equals the leaf's (a key collision is a miss); up to 4 hits per tree and
        launch, independently of set_terminal_extra."""
        assert self.width == 1 or not on, "the evaluation cache needs one simulation in flight per tree"
        self.ctx.bind_stream()
        check(lib().cz_search_set_eval_cache(self.ctx.h, 1 if on else 0), "cz_search_set_eval_cache")
        self.eval_cache = bool(on)
        if not on:
            self.xcache_log2 = 0    # the cross-tree level lives behind the per-tree probe: the library frees it with it

    def set_xcache(self, log2_entries):
        """Cross-tree level of the evaluation cache (cz_search_set_xcache; needs set_eval_cache(True)): one table of
        2**log2_entries self-contained entries (1088 bytes each) shared by all trees of the context — a leaf whose position ANY
        tree has evaluated is expanded inside the select launch from the remembered row (verified against the stored position;
        trees bit-identical).  0 frees it; calling it again empties it (do so whenever the weights change — set_eval_cache(True)
        empties both levels)."""
        assert self.eval_cache or not log2_entries, "set_eval_cache(True) first"
        self.ctx.bind_stream()
        check(lib().cz_search_set_xcache(self.ctx.h, int(log2_entries)), "cz_search_set_xcache")
        self.xcache_log2 = int(log2_entries)

    def xcache_stats(self):
        """-> dict(hits, lookups, written, lost, replaced) of the cross-tree level since it was last emptied: lost = filings that
        found no room (a full bucket of shallower positions, or every swap lost), replaced = entries that took a deeper one's
        place (part of `written`)."""
        a = (C.c_ulonglong * 5)()
        check(lib().cz_search_xcache_stats5(self.ctx.h, a), "cz_search_xcache_stats5")
        return dict(hits=int(a[0]), lookups=int(a[1]), written=int(a[2]), lost=int(a[3]), replaced=int(a[4]))

    def eval_cache_stats(self):
        """-> (hits, lookups) summed over the trees since the cache was turned on."""
        h, n = C.c_ulonglong(0), C.c_ulonglong(0)
        check(lib().cz_search_eval_cache_stats(self.ctx.h, C.byref(h), C.byref(n)), "cz_search_eval_cache_stats")
        return int(h.value), int(n.value)

    def eval_cache_collisions(self):
        """Key matches the cache refused because the stored position differed from the leaf's (taken as misses)."""
        n = C.c_ulonglong(0)
        check(lib().cz_search_eval_cache_collisions(self.ctx.h, C.byref(n)), "cz_search_eval_cache_collisions")
        return int(n.value)

    def set_sim_target(self, target):
        """Trees stop at `target` completed simulations since their last reset / advance (0: no limit)."""
        check(lib().cz_search_set_sim_target(self.ctx.h, int(target)), "cz_search_set_sim_target")

    # -- tree lifecycle ------------------------------------------------------------------------
    def reset(self, boards, side, rr=None):
        self.ctx.bind_stream()   # launches go to torch's CURRENT stream (also under HIP graph capture)
        boards = torch.as_tensor(np.ascontiguousarray(boards, np.uint8) if not torch.is_tensor(boards) else boards).to(self.dev).reshape(-1, NSQ).contiguous()
        side = torch.as_tensor(np.ascontiguousarray(side, np.uint8) if not torch.is_tensor(side) else side).to(self.dev).contiguous()
        G = boards.shape[0]
        rr_t = None
        if rr is not None:
            rr_t = torch.as_tensor(np.ascontiguousarray(rr, np.int32) if not torch.is_tensor(rr) else rr).to(self.dev).to(torch.int32).contiguous()
        check(lib().cz_search_reset(self.ctx.h, _ptr(boards), _ptr(side), _ptr(rr_t), G), "cz_search_reset")
        if self.G != G or self.planes is None:
            self.G = G
            self.planes = torch.zeros((G * self.width, 9, 10, self.channels), dtype=self.plane_dtype, device=self.dev)
            self.need = torch.zeros(G * self.width, dtype=torch.uint8, device=self.dev)
        self._keep = (boards, side, rr_t)

    def reload(self, which, boards, side, rr=None):
        """MCTS_tree.reload / GameBoard.reload for the games that are over (main.py:255-258,604-608): trees with
        which[g] != 0 start afresh from boards[g] / side[g] / rr[g] ([G] arrays), the others are untouched."""
        self.ctx.bind_stream()
        which = self._dev_u8(which)
        boards = self._dev_u8(boards).reshape(-1, NSQ)
        side = self._dev_u8(side)
        assert which.numel() == self.G and boards.shape[0] == self.G and side.numel() == self.G
        rr_t = None
        if rr is not None:
            rr_t = torch.as_tensor(np.ascontiguousarray(rr, np.int32) if not torch.is_tensor(rr) else rr).to(self.dev).to(torch.int32).contiguous()
        check(lib().cz_search_reload(self.ctx.h, _ptr(which), _ptr(boards), _ptr(side), _ptr(rr_t)), "cz_search_reload")

    def _dev_u8(self, x):
        if not torch.is_tensor(x):
            x = torch.from_numpy(np.ascontiguousarray(x).astype(np.uint8))
        return x.to(self.dev).to(torch.uint8).contiguous()

    def select(self, mode=1, active=None, k=None):
        """-> (leaf planes [G*k,9,10,C] device tensor, needs_eval [G*k] u8 device tensor); k <= width descents per
        tree (default: the engine's width)."""
        self.ctx.bind_stream()   # launches go to torch's CURRENT stream (also under HIP graph capture)
        k = self.width if k is None else int(k)
        self._k = k
        act = _mask(active, self.dev)
        if self.width > 1:
            check(lib().cz_search_select_k(self.ctx.h, int(mode), k, _ptr(act), _ptr(self.planes), self._pd, self.channels,
                                           _ptr(self.need)), "cz_search_select_k")
            self._act = act
            return self.planes[:self.G * k], self.need[:self.G * k]
        else:
            check(lib().cz_search_select(self.ctx.h, int(mode), _ptr(act), _ptr(self.planes), self._pd, self.channels,
                                         _ptr(self.need)), "cz_search_select")
        self._act = act
        return self.planes, self.need

    def expand_backup(self, logits, value):
        """logits [G,2086], value [G,1] or [G]: float32 or bfloat16 device tensors."""
        self.ctx.bind_stream()   # launches go to torch's CURRENT stream (also under HIP graph capture)
        if logits.dtype != value.dtype:
            value = value.to(logits.dtype)
        dt = BF16 if logits.dtype == torch.bfloat16 else F32
        if dt == F32 and logits.dtype != torch.float32:
            logits, value = logits.float(), value.float()
        logits = logits.contiguous()
        value = value.contiguous()
        k = getattr(self, "_k", self.width)
        assert logits.shape == (self.G * k, NLABELS) and value.numel() == self.G * k
        if self.width > 1:
            check(lib().cz_search_expand_backup_k(self.ctx.h, k, _ptr(logits), _ptr(value), dt), "cz_search_expand_backup_k")
        else:
            check(lib().cz_search_expand_backup(self.ctx.h, _ptr(logits), _ptr(value), dt), "cz_search_expand_backup")

    def expand_backup_fc(self, z, value, pfc_w, pfc_b, compact=False):
        """expand_backup with the policy FC folded in (cz_search_expand_backup_fc): z [G,90,3] f32 head-conv outputs,
        value [G] or [G,1] f32, pfc_w [2086,180] f32 (torch layout), pfc_b [2086] f32.  Width 1 only.
        compact=True: z / value rows are the ones select_compact handed out (tree g -> row slot_of[g])."""
        self.ctx.bind_stream()   # launches go to torch's CURRENT stream (also under HIP graph capture)
        assert self.width == 1, "expand_backup_fc pairs with the one-simulation-per-tree select"
        assert z.dtype == torch.float32 and z.is_contiguous() and z.shape == (self.G, 90, 3)
        value = value.float().contiguous()
        assert value.numel() == self.G and pfc_w.is_contiguous() and pfc_w.shape == (NLABELS, 180) and pfc_b.numel() == NLABELS
        check(lib().cz_search_expand_backup_fc(self.ctx.h, _ptr(z), _ptr(value), _ptr(pfc_w), _ptr(pfc_b), 1 if compact else 0),
              "cz_search_expand_backup_fc")

    def select_compact(self, mode=1, active=None):
        """select() with compact evaluation batches: the leaf planes of the trees that need a net evaluation go to rows
        0 .. n-1 of the planes buffer (cz_search_select_compact).  Returns (planes, n_rows_ptr): the full planes tensor and
        the DEVICE address of n (an int), to be handed to the net through set_batch_count — no host synchronisation."""
        self.ctx.bind_stream()   # launches go to torch's CURRENT stream (also under HIP graph capture)
        import ctypes as C
        assert self.width == 1
        act = _mask(active, self.dev)
        slot_p, n_p = C.c_void_p(), C.c_void_p()
        check(lib().cz_search_select_compact(self.ctx.h, int(mode), _ptr(act), _ptr(self.planes), self._pd, self.channels,
                                             C.byref(slot_p), C.byref(n_p)), "cz_search_select_compact")
        self._act = act
        return self.planes, n_p

    def eval_totals(self):
        """(rows evaluated, compact steps) since the context was created (synchronises)."""
        self.ctx.bind_stream()   # launches go to torch's CURRENT stream (also under HIP graph capture)
        import ctypes as C
        r, n = C.c_ulonglong(0), C.c_ulonglong(0)
        check(lib().cz_search_eval_totals(self.ctx.h, C.byref(r), C.byref(n)), "cz_search_eval_totals")
        return int(r.value), int(n.value)

    def root_stats(self):
        self.ctx.bind_stream()   # launches go to torch's CURRENT stream (also under HIP graph capture)
        G, dev = self.G, self.dev
        out = dict(label=torch.empty((G, MAXMOVES), dtype=torch.int16, device=dev),
                   N=torch.empty((G, MAXMOVES), dtype=torch.int32, device=dev),
                   Q=torch.empty((G, MAXMOVES), dtype=torch.float32, device=dev),
                   P=torch.empty((G, MAXMOVES), dtype=torch.float32, device=dev),
                   W=torch.empty((G, MAXMOVES), dtype=torch.float32, device=dev),
                   count=torch.empty(G, dtype=torch.int16, device=dev))
        check(lib().cz_search_root_stats(self.ctx.h, _ptr(out["label"]), _ptr(out["N"]), _ptr(out["Q"]), _ptr(out["P"]),
                                         _ptr(out["W"]), _ptr(out["count"])), "cz_search_root_stats")
        return out

    def root_stats_host(self):
        st = self.root_stats()
        return dict(label=st["label"].cpu().numpy().view(np.uint16), N=st["N"].cpu().numpy(), Q=st["Q"].cpu().numpy(),
                    P=st["P"].cpu().numpy(), W=st["W"].cpu().numpy(), count=st["count"].cpu().numpy().view(np.uint16))

    def advance(self, played):
        self.ctx.bind_stream()   # launches go to torch's CURRENT stream (also under HIP graph capture)
        if not torch.is_tensor(played):
            played = torch.from_numpy(np.ascontiguousarray(played, np.uint16).view(np.int16))
        played = played.to(self.dev).contiguous()
        check(lib().cz_search_advance(self.ctx.h, _ptr(played)), "cz_search_advance")

    def advance_ready(self, thr, next_thr, start_boards, start_side, start_rr=None, banked=None, reloaded=None):
        """The greedy driver of a long search loop, all on the device (cz_search_pick_ready -> cz_search_advance ->
        cz_search_reload_finished): every tree that has completed thr[g] simulations (int32 [G] device tensor, set to
        next_thr for the trees that move) plays its most visited root child; games that are then over restart from
        start_boards / start_side / start_rr.  banked / reloaded: int64 device scalars accumulating the simulations of the
        searches closed and the games restarted.  Returns the (played, ready) device tensors of this call."""
        self.ctx.bind_stream()
        G, dev = self.G, self.dev
        if getattr(self, "_ar", None) is None or self._ar[0].numel() != G:
            self._ar = (torch.empty(G, dtype=torch.int16, device=dev), torch.empty(G, dtype=torch.uint8, device=dev))
        played, ready = self._ar
        assert thr.dtype == torch.int32 and thr.is_cuda and thr.numel() == G
        check(lib().cz_search_pick_ready(self.ctx.h, _ptr(thr), int(next_thr), _ptr(played), _ptr(ready), _ptr(banked)), "cz_search_pick_ready")
        check(lib().cz_search_advance(self.ctx.h, _ptr(played)), "cz_search_advance")
        check(lib().cz_search_reload_finished(self.ctx.h, _ptr(ready), _ptr(played), _ptr(start_boards), _ptr(start_side), _ptr(start_rr),
                                              _ptr(reloaded)), "cz_search_reload_finished")
        return played, ready

    def status(self):
        self.ctx.bind_stream()   # launches go to torch's CURRENT stream (also under HIP graph capture)
        G, dev = self.G, self.dev
        st, nodes, sims, depth = (torch.empty(G, dtype=torch.int32, device=dev) for _ in range(4))
        check(lib().cz_search_status(self.ctx.h, _ptr(st), _ptr(nodes), _ptr(sims), _ptr(depth)), "cz_search_status")
        return st, nodes, sims, depth

    def root_state(self):
        self.ctx.bind_stream()   # launches go to torch's CURRENT stream (also under HIP graph capture)
        G, dev = self.G, self.dev
        b = torch.empty((G, NSQ), dtype=torch.uint8, device=dev)
        s = torch.empty(G, dtype=torch.uint8, device=dev)
        rr = torch.empty(G, dtype=torch.int32, device=dev)
        check(lib().cz_search_root_state(self.ctx.h, _ptr(b), _ptr(s), _ptr(rr)), "cz_search_root_state")
        return b, s, rr

    def tree_dump(self, g, max_records=1 << 20):
        self.ctx.bind_stream()   # launches go to torch's CURRENT stream (also under HIP graph capture)
        out = np.zeros((max_records, 7), np.int32)
        n = lib().cz_search_tree_dump(self.ctx.h, int(g), out.ctypes.data_as(C.c_void_p), int(max_records))
        if n < 0:
            check(n, "cz_search_tree_dump")
        if n > max_records:
            raise _lib.CchessHipError("tree_dump: %d records > %d" % (n, max_records))
        return out[:n].copy()

    # -- the hot loop ------------------------------------------------------------------------------
    def step(self, forward, mode=1, active=None, tap=None, pre_net=None):
        """One lock-step simulation for every tree: select -> forward(planes) -> expand_backup.
        `forward` maps the device planes tensor to (logits [G,2086], value [G,1]) device tensors; if it is a
        PolicyValueNet with the fused hip backend (has `search_eval`) and width is 1, the policy FC is evaluated
        inside the expansion for the legal moves only (expand_backup_fc) and no logits tensor exists at all.
        tap (parity tests, bench events): called as tap(planes, z_or_logits, value) between the net and the expansion —
        the tensors are the ones the expansion kernel is about to read; pre_net() is called between select and the net."""
        net = getattr(forward, "__self__", forward)
        if self.width == 1 and getattr(net, "search_eval", None) is not None and net.fused_search:
            if self.compact:   # terminal / drawn / parked trees cost the net nothing
                planes, n_rows = self.select_compact(mode, active)
                if pre_net is not None:
                    pre_net()
                z, value = net.search_eval(planes, n_rows)
                if tap is not None:
                    tap(planes, z, value)
                self.expand_backup_fc(z, value, net.pfc_w_rows, net.pfc_b_f32, compact=True)
            else:
                planes, _ = self.select(mode, active)
                if pre_net is not None:
                    pre_net()
                z, value = net.search_eval(planes)
                if tap is not None:
                    tap(planes, z, value)
                self.expand_backup_fc(z, value, net.pfc_w_rows, net.pfc_b_f32)
            return
        planes, _ = self.select(mode, active)
        if pre_net is not None:
            pre_net()
        logits, value = forward(planes)
        if tap is not None:
            tap(planes, logits, value)
        self.expand_backup(logits, value)

    def _active_bool(self, active):
        """The active mask as a bool device tensor for the host-side liveness tests of search(): None, a tensor / array,
        or a raw device pointer (what SelfPlay hands over in parking mode: cz_selfplay_active) — that one is downloaded."""
        if active is None:
            return None
        if isinstance(active, C.c_void_p):
            buf = np.empty(self.G, np.uint8)
            check(lib().cz_download(self.ctx.h, buf.ctypes.data_as(C.c_void_p), active, self.G), "cz_download")
            return torch.from_numpy(buf).to(self.dev).bool()
        return torch.as_tensor(active).to(self.dev).bool()

    def search(self, forward, playouts, active=None, root_done=False):
        """MCTS_tree.main (main.py:473-493) for all trees: expand unexpanded roots, then EXACTLY `playouts` simulations
        per (active, unparked) tree.  With width k > 1 up to k simulations are in flight per tree and step; a descent
        that runs into a pending expansion is abandoned (see k_select_k), so a step can complete fewer than k — the
        per-tree budget (cz_search_set_sim_target) and a few extra steps make up for it.  Returns the number of steps.
        root_done=True: the caller has already run the mode-0 step (root expansion).
        (Round 6 replayed the k-wide lock-step as one captured HIP graph for the --mode play shape — one tree, 16 simulations in
        flight — and measured no gain, 0.407 against 0.410 ms per lock-step: the step is five SERIAL small launches on the GPU, a
        203 us trunk for 16 positions and a k-wide select / expand that walk their 16 descents one after the other in one wave,
        not host time; the capture was removed again.)"""
        playouts = int(playouts)
        if not root_done:
            self.step(forward, mode=0, active=active)
        if self.width == 1 and self.terminal_extra == 0:
            for _ in range(playouts):
                self.step(forward, mode=1, active=active)
            return playouts
        if self.width == 1:
            # terminal simulations complete inside the select launches: a tree needs one lock-step per simulation that
            # needs the net, so the search is over when every tree has counted `playouts` (checked every 32 steps)
            base = self.status()[2]
            act = self._active_bool(active)   # parked / inactive slots neither decide the schedule nor keep the loop alive
            if act is not None:
                if not bool(act.any().item()):
                    return 0
                base = base[act]
            if not bool((base == base[0]).all().item()):   # stacked searches on unequal counters: the plain schedule
                extra = self.terminal_extra
                self.set_terminal_extra(0)
                try:
                    for _ in range(playouts):
                        self.step(forward, mode=1, active=active)
                finally:
                    self.set_terminal_extra(extra)
                return playouts
            target = int(base[0].item()) + playouts
            steps = 0
            self.set_sim_target(target)
            try:
                while steps < playouts:
                    for _ in range(min(32, playouts - steps)):
                        self.step(forward, mode=1, active=active)
                        steps += 1
                    st, _, sims, _ = self.status()
                    live = (st & ~8) == 0
                    if act is not None:
                        live = live & act
                    if not bool(((sims < target) & live).any().item()):
                        break
            finally:
                self.set_sim_target(0)
            return steps
        base = self.status()[2].clone()      # searches may be stacked on one root: the target is relative
        check(lib().cz_search_set_sim_target(self.ctx.h, 0), "cz_search_set_sim_target")
        # per-tree targets differ only if `base` does; the kernel takes one number, so stacked searches on trees with
        # different counters fall back to the unbudgeted schedule for the bulk and finish one simulation at a time
        uniform = bool((base == base[0]).all().item())
        target = int(base[0].item()) + playouts
        steps = 0
        try:
            if uniform:
                check(lib().cz_search_set_sim_target(self.ctx.h, target), "cz_search_set_sim_target")
            one = lambda: self.step(forward, mode=1, active=active)
            n = (playouts + self.width - 1) // self.width
            for _ in range(n):
                one()
            steps = n
            if uniform:
                # the shortfall of abandoned descents: usually 1-3 steps — but behind a FRESH root every descent of a step picks the
                # same child (the reference never updates the root's N, quirk Q2: U = 0 there, and its virtual loss leaves Q alone),
                # so all but one are abandoned and a step completes ONE simulation: up to `playouts` steps, like the reference's
                # sixteen coroutines queueing behind one expansion (round 6: the cap of 4 n + 8 extra steps ended such a search short)
                act = self._active_bool(active)
                while steps < 5 * playouts + 8:
                    st, _, sims, _ = self.status()
                    live = (st & ~8) == 0
                    if act is not None:
                        live = live & act
                    if not bool(((sims < target) & live).any().item()):
                        break
                    one()
                    steps += 1
        finally:
            check(lib().cz_search_set_sim_target(self.ctx.h, 0), "cz_search_set_sim_target")
        return steps
