#coding:utf-8
"""Drop-in for the reference's main.py on the MI355X hot path: same CLI (--mode train/play and every
flag of main.py:1557-1575), same public names (GameBoard, MCTS_tree, leaf_node, cchess_main,
labels_array, label2i, flipped_uci_labels, is_kill_move, softmax ...), but every rule / search
operation is a HIP kernel behind the C-ABI of include/cchess_hip.h and the net is the PyTorch-ROCm
re-expression with the fused MFMA tower.  Nothing here falls back to a CPU implementation: without
the HIP library or a GPU the constructors raise.

Where the reference is single-game Python, this façade keeps the one-game call shapes
(get_legal_moves(state, player) -> list[str], MCTS_tree.main(...), cchess_main.selfplay() ...) by
running the batched kernels with G = 1; `cchess_main.run()` — the training loop — plays `--games`
games in lock-step per GPU (extra flag, default 256), which is where the throughput is.

Semantics notes: with search_threads = 1 the device search is bit-identical to the
reference (whole trees, see tests/golden).  search_threads = k > 1 keeps k simulations in flight per tree
with the reference's virtual loss (N += 3, W -= 3), batched deterministically (the reference's own
interleaving depends on wall-clock asyncio sleeps); the batched `run()` loop uses one simulation in
flight per tree because thousands of trees already fill the net batch.  Pseudo-legal rules, raw-logit
priors, the root never being backed up, and the 9-stride plane quirk are all reproduced.
"""
import argparse
import json
import os
import random
import sys
import time
from collections import defaultdict, deque

import numpy as np

from cchess_zero_amd import _lib
from cchess_zero_amd.notation import (START_STATE, board_to_state, player_to_side, side_to_player, state_to_board)

# ---- move vocabulary (main.py:23-65, 208-217 of the reference) ------------------------------------------
_T = None


def _tables():
    global _T
    if _T is None:
        _T = _lib.tables()
    return _T


def create_uci_labels():
    return list(_tables()["labels"])


def flipped_uci_labels(param):
    def repl(x):
        return "".join([(str(9 - int(a)) if a.isdigit() else a) for a in x])
    return [repl(x) for x in param]


pieces_order = 'KARBNPCkarbnpc'
ind = {pieces_order[i]: i for i in range(14)}
labels_array = create_uci_labels()
labels_len = len(labels_array)
unflipped_index = [int(i) for i in _tables()["unflip"]]
i2label = {i: val for i, val in enumerate(labels_array)}
label2i = {val: i for i, val in enumerate(labels_array)}
c_PUCT = 5
virtual_loss = 3


def get_pieces_count(state):
    return sum(1 for s in state if s.isalpha())


def is_kill_move(state_prev, state_next):
    return get_pieces_count(state_prev) - get_pieces_count(state_next)


def softmax(x):
    probs = np.exp(x - np.max(x))
    probs /= np.sum(probs)
    return probs


# ---- device plumbing shared by the façade classes ------------------------------------------------------
_RULES = None


def _rules():
    global _RULES
    if _RULES is None:
        from cchess_zero_amd.rules import Rules
        _RULES = Rules()
    return _RULES


class GameBoard(object):
    """Reference: main.py:579-1109.  State strings in, state strings / move lists out; the work is done by
    the K1/K2 kernels (batch of one)."""
    Ny = 10
    Nx = 9

    def __init__(self):
        self.reload()

    def reload(self):
        self.state = START_STATE
        self.round = 1
        self.current_player = "w"
        self.restrict_round = 0

    @staticmethod
    def get_legal_moves(state, current_player):
        r = _rules()
        moves, count, _ = r.movegen(state_to_board(state)[None], np.array([player_to_side(current_player)], np.uint8), want_mask=False)
        n = int(count.cpu().numpy().view(np.uint16)[0])
        if n == 0xFFFF:
            raise RuntimeError("move generation overflow (>128 moves)")
        lab = moves[0, :n].cpu().numpy().view(np.uint16)
        return [labels_array[i] for i in lab]

    @staticmethod
    def sim_do_action(in_action, in_state):
        import torch
        r = _rules()
        b = torch.from_numpy(state_to_board(in_state)[None].copy()).to(r.dev)
        s = torch.zeros(1, dtype=torch.uint8, device=r.dev)
        r.apply_move(b, s, np.array([label2i[in_action]], np.uint16).view(np.int16))
        return board_to_state(b[0].cpu().numpy())

    @staticmethod
    def board_to_pos_name(board):
        for d in "23456789":
            board = board.replace(d, "1" * int(d))
        return board.split("/")

    @staticmethod
    def check_bounds(toY, toX):
        return 0 <= toY < GameBoard.Ny and 0 <= toX < GameBoard.Nx

    @staticmethod
    def print_borad(board, action=None):
        rows = GameBoard.board_to_pos_name(board)
        print("  abcdefghi")
        for i, line in enumerate(rows):
            line = line.replace("1", " ")
            if action is not None and i == int(action[1]):
                x = "abcdefghi".index(action[0])
                line = line[:x] + "x" + line[x + 1:]
            print(i, line)


class leaf_node(object):
    """Read-only view of one root child (reference: main.py:93-206; the tree itself lives on the device)."""

    def __init__(self, in_parent, in_prior_p, in_state, N=0, Q=0.0, W=0.0):
        self.P = in_prior_p
        self.Q = Q
        self.N = N
        self.W = W
        self.v = 0
        self.U = 0
        self.parent = in_parent
        self.child = {}
        self.state = in_state

    def is_leaf(self):
        return self.child == {}


class _RootView(object):
    def __init__(self, tree):
        self._t = tree
        self.parent = None

    @property
    def child(self):
        return self._t._children()

    @property
    def state(self):
        return self._t._state


class MCTS_tree(object):
    """Reference: main.py:234-577.  One device tree (G = 1) driven in lock-step: select -> forward -> expand/backup."""

    def __init__(self, in_state, in_forward, search_threads):
        from cchess_zero_amd.engine import SearchEngine
        self.noise_eps = 0.25
        self.dirichlet_alpha = 0.3
        self.p_ = 1.0
        self.c_puct = 5
        self.virtual_loss = 3
        self.forward = in_forward
        self.search_threads = search_threads
        self._cap = int(os.environ.get("CCHESS_TREE_NODES", 400000))
        # search_threads coroutines of the reference = that many simulations in flight with virtual loss
        self._width = max(1, min(int(search_threads), 64))
        self._eng = SearchEngine(1, self._cap, width=self._width)
        self._state = None
        self._player = None
        self._rr = 0
        self._fresh = True
        self._set_root(in_state, "w", 0)
        self.root = _RootView(self)

    # -- helpers -------------------------------------------------------------------------------------
    def _set_root(self, state, player, rr):
        self._eng.reset(state_to_board(state)[None], np.array([player_to_side(player)], np.uint8), np.array([rr], np.int32))
        self._state, self._player, self._rr = state, player, rr
        self._cache = None

    def _children(self):
        if self._cache is None:
            st = self._eng.root_stats_host()
            n = int(st["count"][0])
            d = {}
            for i in range(n):
                mv = labels_array[int(st["label"][0, i])]
                d[mv] = leaf_node(self.root, float(st["P"][0, i]), None, int(st["N"][0, i]), float(st["Q"][0, i]), float(st["W"][0, i]))
            self._cache = d
        return self._cache

    def _device_forward(self):
        """If `forward` is the bound method of our network, skip the host round trip."""
        owner = getattr(self.forward, "__self__", None)
        return getattr(owner, "forward_device", None)

    def _step(self, mode, k=None):
        import torch
        planes, need = self._eng.select(mode, k=k)
        fd = self._device_forward()
        if fd is not None:
            logits, value = fd(planes)
        else:
            lg, v = self.forward(planes.cpu().numpy())
            logits = torch.from_numpy(np.ascontiguousarray(lg, np.float32)).to(planes.device)
            value = torch.from_numpy(np.ascontiguousarray(v, np.float32)).to(planes.device)
        self._eng.expand_backup(logits, value)

    # -- reference surface ------------------------------------------------------------------------------
    def reload(self):
        self._set_root(START_STATE, "w", 0)

    def is_black_turn(self, current_player):
        return current_player == 'b'

    def main(self, state, current_player, restrict_round, playouts):
        # the reference passes restrict_round into every search (main.py:1337); the device tree carries it in its root, so a
        # caller that changes it behind the tree's back (a fresh game set up at another count) gets a fresh root
        if state != self._state or current_player != self._player or int(restrict_round) != int(self._rr):
            self._set_root(state, current_player, restrict_round)
        self._step(0)                       # root expansion if needed (main.py:475-487)
        if self._width == 1:
            for _ in range(int(playouts)):
                self._step(1)               # one simulation (main.py:350-435)
        elif self._device_forward() is not None and not os.environ.get("CCHESS_PLAY_EAGER"):
            # up to `search_threads` simulations in flight, exactly `playouts` in total (the per-tree budget of SearchEngine.search:
            # one status read-back per step instead of two, no host-side bookkeeping of k)
            self._eng.search(self._device_forward(), int(playouts), root_done=True)
        else:                               # a host `forward`: one host round trip per step, counted on the host
            base = int(self._eng.status()[2].cpu().numpy()[0])
            done, stall = 0, 0
            while done < int(playouts) and stall < 4:
                self._step(1, k=min(self._width, int(playouts) - done))
                now = int(self._eng.status()[2].cpu().numpy()[0]) - base
                stall = stall + 1 if now == done else 0
                done = now
        self._cache = None
        status = int(self._eng.status()[0].cpu().numpy()[0])
        if status & 1:
            raise MemoryError("tree node pool exhausted; raise CCHESS_TREE_NODES")
        if status & 2:
            raise ValueError("max() arg is an empty sequence")   # what the reference raises (quirk Q7)

    def Q(self, move) -> float:
        c = self._children()
        if move not in c:
            print("{} not exist in the child".format(move))
            return 0.0
        return c[move].Q

    def update_tree(self, act):
        if act not in self._children():
            raise KeyError(act)
        nxt = GameBoard.sim_do_action(act, self._state)
        self._rr = self._rr + 1 if is_kill_move(self._state, nxt) == 0 else 0
        self._eng.advance(np.array([label2i[act]], np.uint16))
        self._state = nxt
        self._player = "w" if self._player == "b" else "b"
        self._cache = None

    def try_flip(self, state, current_player, flip=False):
        if not flip:
            return state, current_player
        rows = state.split('/')
        return "/".join(r.swapcase() for r in reversed(rows)), ('w' if current_player == 'b' else 'b')

    def state_to_positions(self, state):
        """[9,10,14] planes of an (already canonical) state string, via the K3 kernel."""
        r = _rules()
        return r.encode_planes(state_to_board(state)[None], np.zeros(1, np.uint8)).cpu().numpy()[0]

    def generate_inputs(self, in_state, current_player):
        r = _rules()
        return r.encode_planes(state_to_board(in_state)[None], np.array([player_to_side(current_player)], np.uint8)).cpu().numpy()[0]


class cchess_main(object):
    """Reference: main.py:1118-1554."""

    def __init__(self, playout=400, in_batch_size=128, exploration=True, in_search_threads=16, processor="cpu",
                 num_gpus=1, res_block_nums=7, human_color='b', games=256):
        from policy_value_network import policy_value_network
        from policy_value_network_gpus import policy_value_network_gpus
        self.epochs = 5
        self.playout_counts = playout
        self.temperature = 1
        self.batch_size = in_batch_size
        self.game_batch = 400
        self.top_steps = 30
        self.top_temperature = 1
        self.eta = 0.03
        self.learning_rate = 0.001
        self.lr_multiplier = 1.0
        self.buffer_size = 10000
        self.data_buffer = deque(maxlen=self.buffer_size)
        self.game_borad = GameBoard()
        # `--processor cpu` selected the single-device TF graph in the reference (main.py:1142); both
        # settings run on the local MI355X here, `gpu` additionally honours the torch.distributed launcher.
        self.policy_value_netowrk = policy_value_network(res_block_nums) if processor == 'cpu' else policy_value_network_gpus(num_gpus, res_block_nums)
        self.search_threads = in_search_threads
        self.mcts = MCTS_tree(self.game_borad.state, self.policy_value_netowrk.forward, self.search_threads)
        self.exploration = exploration
        self.resign_threshold = -0.8
        self.global_step = 0
        self.kl_targ = 0.025
        self.log_file = open(os.path.join(os.getcwd(), 'log_file.txt'), 'w')
        self.human_color = human_color
        self.games = games
        self.num_gpus = num_gpus
        self.update_seed = 20260925   # mini-batch / shuffle seed shared by all ranks (rank-consistent control flow)

    @staticmethod
    def flip_policy(prob):
        prob = np.asarray(prob).flatten()
        return prob[np.asarray(unflipped_index)]

    # ---- one move of one game (main.py:1332-1358) -----------------------------------------------------
    def get_action(self, state, temperature=1e-3):
        self.mcts.main(state, self.game_borad.current_player, self.game_borad.restrict_round, self.playout_counts)
        actions_visits = [(act, nod.N) for act, nod in self.mcts.root.child.items()]
        actions, visits = zip(*actions_visits)
        with np.errstate(divide="ignore"):
            probs = softmax(1.0 / temperature * np.log(visits))
        move_probs = [[actions, probs]]
        if self.exploration:
            act = np.random.choice(actions, p=0.75 * probs + 0.25 * np.random.dirichlet(0.3 * np.ones(len(probs))))
        else:
            act = np.random.choice(actions, p=probs)
        win_rate = self.mcts.Q(act)
        self.mcts.update_tree(act)
        return act, move_probs, win_rate

    def check_end(self):
        s = self.game_borad.state
        if s.find('K') == -1 or s.find('k') == -1:
            if s.find('K') == -1:
                print("Green is Winner")
                return True, "b"
            print("Red is Winner")
            return True, "w"
        elif self.game_borad.restrict_round >= 60:
            print("TIE! No Winners!")
            return True, "t"
        return False, ""

    def _advance_board(self, action):
        gb = self.game_borad
        last_state = gb.state
        gb.state = GameBoard.sim_do_action(action, gb.state)
        gb.round += 1
        gb.current_player = "w" if gb.current_player == "b" else "b"
        gb.restrict_round = gb.restrict_round + 1 if is_kill_move(last_state, gb.state) == 0 else 0

    def _net_move_probs(self):
        """`--ai_function net` branch of select_move / get_hint (main.py:1300-1324, 1437-1461)."""
        gb = self.game_borad
        positions = np.expand_dims(self.mcts.generate_inputs(gb.state, gb.current_player), 0)
        action_probs, value = self.mcts.forward(positions)
        if self.mcts.is_black_turn(gb.current_player):
            action_probs = cchess_main.flip_policy(action_probs)
        action_probs = np.asarray(action_probs).flatten()
        moves = GameBoard.get_legal_moves(gb.state, gb.current_player)
        tot_p = 1e-8
        d = defaultdict(float)
        for a in moves:
            d[a] = action_probs[label2i[a]]
            tot_p += d[a]
        for a in d:
            d[a] /= tot_p
        return d, float(value[0, 0])

    def select_move(self, mcts_or_net):
        if mcts_or_net == "mcts":
            action, probs, win_rate = self.get_action(self.game_borad.state, self.temperature)
        else:
            d, win_rate = self._net_move_probs()
            action = max(d.items(), key=lambda kv: kv[1])[0]
        print('Win rate for player {} is {:.4f}'.format(self.game_borad.current_player, win_rate))
        print(self.game_borad.current_player, " now take a action : ", action, "[Step {}]".format(self.game_borad.round))
        self._advance_board(action)
        self.game_borad.print_borad(self.game_borad.state)
        if self.human_color == 'w':
            action = "".join(flipped_uci_labels(action))
        sx, sy, dx, dy = "abcdefghi".index(action[0]), int(action[1]), "abcdefghi".index(action[2]), int(action[3])
        return (sx, sy, dx - sx, dy - sy), win_rate

    def human_move(self, coord, mcts_or_net):
        win_rate = 0
        action = "abcdefghi"[coord[0]] + str(coord[1]) + "abcdefghi"[coord[2]] + str(coord[3])
        if self.human_color == 'w':
            action = "".join(flipped_uci_labels(action))
        if mcts_or_net == "mcts":
            if self.mcts.root.child == {}:
                self.mcts.main(self.game_borad.state, self.game_borad.current_player, self.game_borad.restrict_round, self.playout_counts)
            win_rate = self.mcts.Q(action)
            self.mcts.update_tree(action)
        self._advance_board(action)
        return win_rate

    def get_hint(self, mcts_or_net, reverse, disp_mcts_msg_handler):
        if mcts_or_net == "mcts":
            if self.mcts.root.child == {}:
                disp_mcts_msg_handler()
                self.mcts.main(self.game_borad.state, self.game_borad.current_player, self.game_borad.restrict_round, self.playout_counts)
            actions, visits = zip(*[(a, n.N) for a, n in self.mcts.root.child.items()])
            with np.errstate(divide="ignore"):
                probs = softmax(1.0 / self.temperature * np.log(visits))
            d = defaultdict(float)
            for a, p in zip(actions, probs):
                d["".join(flipped_uci_labels(a)) if self.human_color == 'w' else a] = p
        else:
            dd, _ = self._net_move_probs()
            d = defaultdict(float)
            for a, p in dd.items():
                d["".join(flipped_uci_labels(a)) if self.human_color == 'w' else a] = p
        return sorted(d.items(), key=lambda item: item[1], reverse=reverse)

    # ---- one self-play game through the single-tree surface (main.py:1493-1554) -------------------------
    def selfplay(self):
        self.game_borad.reload()
        self.mcts.reload()
        states, mcts_probs, current_players = [], [], []
        z = None
        game_over = False
        start_time = time.time()
        while not game_over:
            gb = self.game_borad
            player = gb.current_player
            state_before = gb.state
            action, probs, win_rate = self.get_action(gb.state, self.temperature)
            state, _ = self.mcts.try_flip(state_before, player, self.mcts.is_black_turn(player))
            states.append(state)
            prob = np.zeros(labels_len)
            for a, p in zip(probs[0][0], probs[0][1]):
                if self.mcts.is_black_turn(player):
                    a = "".join((str(9 - int(c)) if c.isdigit() else c) for c in a)
                prob[label2i[a]] = p
            mcts_probs.append(prob)
            current_players.append(player)
            self._advance_board(action)
            s = self.game_borad.state
            if s.find('K') == -1 or s.find('k') == -1:
                winner = "b" if s.find('K') == -1 else "w"
                z = np.where(np.array(current_players) == winner, 1.0, -1.0)
                game_over = True
                print("Game end. Winner is player : ", winner, " In {} steps".format(self.game_borad.round - 1))
            elif self.game_borad.restrict_round >= 60:
                z = np.zeros(len(current_players))
                game_over = True
                print("Game end. Tie in {} steps".format(self.game_borad.round - 1))
        self.mcts.reload()
        print("Using time {} s".format(time.time() - start_time))
        return zip(states, mcts_probs, z), len(z)

    # ---- the batched training loop (main.py:1157-1248) ------------------------------------------------------
    def selfplay_batch(self, games=None, max_plies=None, target_games=None):
        """Self-play on `games` game slots of this GPU, device-resident and continuous (a finished game's slot starts
        the next game at once, cchess_zero_amd/selfplay.py) until `target_games` more games (default: one per slot) have
        finished; returns the packed (s, pi, z) records of all ranks (one all-gather).  max_plies bounds the number of
        plies played (tests).

        The game slots LIVE ACROSS BATCHES: a game still in progress when the batch has its `target_games` is neither
        thrown away nor restarted — it goes on in the next batch (with the new weights from there on; the evaluation cache
        is emptied when the weights change) and its records are drained when it ends.  So every game that is started
        reaches the buffer, long games and 60-ply draws included, as in the reference, which plays every game to its end
        (main.py:1228-1240); stopping all slots at a batch boundary would keep only the games shorter than the batch."""
        import torch
        from cchess_zero_amd import parallel
        from cchess_zero_amd.engine import SearchEngine
        from cchess_zero_amd.selfplay import SelfPlay
        G = games or self.games
        target = target_games or G
        # a ply adds ~40 nodes per simulation on top of the subtree kept from the previous ply; a tree that fills its pool
        # stops expanding for the rest of that ply (the move is chosen from the visits it has) and gets its room back when
        # cz_search_advance compacts it
        cap = max(4096, (self.playout_counts + 2) * 128)
        net = self.policy_value_netowrk.net
        sp = getattr(self, "_sp", None)
        if sp is None or sp.eng.G != G or sp.eng.ctx.cap < cap or sp.playouts != self.playout_counts or sp.net is not net:
            # planes are written by the select kernel straight in the fused net's input format (16 channels of its 16-bit type)
            fused = (net.backend == "hip" or net.strict_auto) and net.dtype in (torch.float16, torch.bfloat16)
            eng = SearchEngine(G, cap, torch.cuda.current_device(), plane_dtype=net.dtype if fused else torch.float32,
                               channels=16 if fused else 14)
            # every rank seeds its games differently but reproducibly from the shared Python RNG state
            base_seed = random.randrange(1 << 30)
            rank = int(os.environ.get("RANK", "0"))
            sp = SelfPlay(eng, net, self.playout_counts, self.exploration, self.temperature,
                          seed=base_seed + 7919 * rank, continuous=True,
                          # the evaluation cache pays from a few hundred playouts per move on (in-tree repeat rate 1 % at 100
                          # playouts, 4 % at 400, 12 % at 1600; the lookup costs the select launch 10-25 us)
                          eval_cache=self.playout_counts >= 400,
                          # ... and its cross-tree level: every game starts from the same position, the openings are shared.  Sized
                          # for the 288 GB of the MI355X: ~2 048 entries of 1 088 bytes per game slot (256 games: 2**19 = 0.6 GB,
                          # 8192 games: 2**24 = 18 GB) — measured at 8192 games x 1600 playouts over 48 000 lock-steps
                          # (profiles/r05j_*, r05o_*): no cache 3.61 M simulations/s, per-tree level 3.96 M, 2**22 entries
                          # 4.16 M, 2**24 entries 4.38 M (+21.6 %); the write-once table is full either way
                          xcache_log2=min(24, max(18, (G * 2048 - 1).bit_length())) if self.playout_counts >= 400 else 0,
                          # a drain interval can end every game of every slot in the worst case: room for ~160 plies per slot
                          ring_records=max(65536, 160 * G))
            b0 = np.tile(state_to_board(START_STATE), (G, 1))
            sp.start(b0, np.zeros(G, np.uint8), np.zeros(G, np.int32))
            self._sp, self._batch_eng = sp, eng
            self._sp_weights_step = self.policy_value_netowrk.global_step
        elif self._sp_weights_step != self.policy_value_netowrk.global_step:
            if sp.eval_cache:
                sp.eng.set_eval_cache(True)      # new weights: remembered evaluations are stale (turning it on empties both levels)
            self._sp_weights_step = self.policy_value_netowrk.global_step
        before = sp.stats()
        # asynchronous plies: every game moves when ITS search has had its playouts (simulations that end on a king capture
        # or the 60-ply rule complete inside the select launch and use no net row, so searches differ in length)
        chunks, plies = [], 0
        steps_per_ply = self.playout_counts + 1
        every = max(1, min(8, self.playout_counts // 16))
        while True:
            n = 8 if max_plies is None else max(1, min(8, max_plies - plies))
            sp.run_async(n * steps_per_ply, every=every, terminal_extra=4)
            plies += n
            chunks.append(sp.drain_device(on_overflow="skip").clone())   # an overflowed interval is dropped with a warning, never handed out
            st = sp.stats()
            if st["games"] - before["games"] >= target or (max_plies is not None and plies >= max_plies):
                break
        self.last_selfplay_stats = {k: (st[k] - before[k] if k in ("games", "red_wins", "black_wins", "draws", "plies", "sims", "lock_steps") else st[k]) for k in st}
        self.last_selfplay_sims = self.last_selfplay_stats["sims"]
        rec = torch.cat(chunks, 0) if chunks else sp.ring[:0]
        return parallel.gather_records(rec)

    def policy_update(self, save=True):
        from cchess_zero_amd.train import policy_update
        self.lr_multiplier, info = policy_update(self.policy_value_netowrk, self.data_buffer, self.batch_size, self.epochs,
                                                 self.learning_rate, self.lr_multiplier, self.kl_targ, seed=self.update_seed,
                                                 temperature=self.temperature, save=save)
        self.global_step = self.policy_value_netowrk.global_step
        msg = "kl:{:.5f},lr_multiplier:{:.3f},loss:{},accuracy:{},explained_var_old:{:.3f},explained_var_new:{:.3f}".format(
            info["kl"], self.lr_multiplier, info["loss"], info["accuracy"], info["explained_var_old"], info["explained_var_new"])
        print(msg)
        self.log_file.write(msg + '\n')
        self.log_file.flush()
        return info

    def run(self, max_batches=None):
        """The training loop (main.py:1206-1248).  One iteration = a self-play batch of `games` slots per GPU (thousands
        of samples where the reference's single game gives ~100), so the reference's cadence is rescaled: the buffer
        holds at least the last two batches (the reference: 10 000 samples = ~100 games), the gathered records are
        shuffled before they enter it (so that a bounded buffer is not biased towards the last ranks' games), and the
        number of policy updates per batch grows with the number of new samples (the reference: one update of
        `batch_size` samples x `epochs` per ~100 new samples; here one per `batch_size` new samples, at most 64)."""
        batch_iter = 0
        try:
            while max_batches is None or batch_iter < max_batches:
                batch_iter += 1
                t0 = time.time()
                rec = self.selfplay_batch()
                dt = time.time() - t0
                st = self.last_selfplay_stats
                n = len(rec)
                print("batch i:{}, game slots:{}, games finished:{}, samples:{}, sims/s:{:.0f}".format(
                    batch_iter, self.games, st["games"], n, self.last_selfplay_sims / max(dt, 1e-9)))
                if self.data_buffer.maxlen < 2 * n:
                    self.data_buffer = deque(self.data_buffer, maxlen=2 * n)
                # the buffer keeps the PACKED records (608 B each; the reference's dense tuple is 22 KB): policy_update
                # expands the mini-batch it draws (cchess_zero_amd/train.py)
                order = np.random.RandomState(self.update_seed + batch_iter).permutation(n)   # same on every rank
                self.data_buffer.extend(rec[i] for i in order)
                t1 = time.time()
                updates = 0
                if len(self.data_buffer) > self.batch_size:
                    updates = max(1, min(64, n // self.batch_size))
                    for u in range(updates):   # the reference saves after its one update per game (main.py:1188): once per batch here
                        self.policy_update(save=(u == updates - 1))
                # wall-clock of the product's default loop: one line per batch, also in the log file
                rep = self.policy_value_netowrk.strict_report() if hasattr(self.policy_value_netowrk, "strict_report") else None
                timing = {"batch": batch_iter, "game_slots": self.games, "playout": self.playout_counts, "games_finished": int(st["games"]),
                          "samples_all_ranks": int(n), "simulations": int(self.last_selfplay_sims), "lock_steps": int(st.get("lock_steps", 0)),
                          "selfplay_seconds": round(dt, 3), "sims_per_s": round(self.last_selfplay_sims / max(dt, 1e-9), 1),
                          "policy_updates": updates, "policy_update_seconds": round(time.time() - t1, 3),
                          "net_engine": getattr(self.policy_value_netowrk.net, "engine_name", None), "strict_check": rep,
                          "eval_cache": bool(getattr(self._sp, "eval_cache", False)), "xcache_log2": int(getattr(self._sp, "xcache_log2", 0))}
                msg = "batch_timing: " + json.dumps(timing)
                print(msg, flush=True)
                self.log_file.write(msg + '\n')
                self.log_file.flush()
        except KeyboardInterrupt:
            self.log_file.close()
            self.policy_value_netowrk.save(self.global_step)


def _play_headless(args):
    """`--mode play` without tkinter: AI vs AI (ai_count 2) or human-vs-AI over stdin (moves like 'b2e2')."""
    m = cchess_main(args.play_playout, args.batch_size, False, args.search_threads, args.processor, args.num_gpus,
                    args.res_block_nums, args.human_color)
    human_turn = (args.ai_count == 1)
    GameBoard.print_borad(m.game_borad.state)
    while True:
        end, who = m.check_end()
        if end:
            return who
        human_to_move = human_turn and ((m.game_borad.current_player == 'w') == (args.human_color == 'w'))
        if human_to_move:
            mv = input("your move (e.g. b2e2): ").strip()
            if mv not in GameBoard.get_legal_moves(m.game_borad.state, m.game_borad.current_player):
                print("illegal move")
                continue
            if args.ai_function == "mcts":
                if m.mcts.root.child == {}:
                    m.mcts.main(m.game_borad.state, m.game_borad.current_player, m.game_borad.restrict_round, m.playout_counts)
                m.mcts.update_tree(mv)
            m._advance_board(mv)
        else:
            m.select_move(args.ai_function)
        time.sleep(max(0.0, args.delay if args.ai_count == 2 else 0.0) * 0)


if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument('--mode', default='train', choices=['train', 'play'], type=str, help='train or play')
    parser.add_argument('--ai_count', default=1, choices=[1, 2], type=int, help='choose ai player count')
    parser.add_argument('--ai_function', default='mcts', choices=['mcts', 'net'], type=str, help='mcts or net')
    parser.add_argument('--train_playout', default=400, type=int, help='mcts train playout')
    parser.add_argument('--batch_size', default=512, type=int, help='train batch_size')
    parser.add_argument('--play_playout', default=400, type=int, help='mcts play playout')
    parser.add_argument('--delay', dest='delay', action='store', nargs='?', default=3, type=float, required=False,
                        help='Set how many seconds you want to delay after each move')
    parser.add_argument('--end_delay', dest='end_delay', action='store', nargs='?', default=3, type=float, required=False,
                        help='Set how many seconds you want to delay after the end of game')
    parser.add_argument('--search_threads', default=16, type=int, help='search_threads')
    parser.add_argument('--processor', default='cpu', choices=['cpu', 'gpu'], type=str, help='cpu or gpu')
    parser.add_argument('--num_gpus', default=1, type=int, help='gpu counts')
    parser.add_argument('--res_block_nums', default=7, type=int, help='res_block_nums')
    parser.add_argument('--human_color', default='b', choices=['w', 'b'], type=str, help='w or b')
    # additions (not in the reference): size of the lock-step game pool per GPU, bounded runs for scripts
    parser.add_argument('--games', default=256, type=int, help='parallel self-play games per GPU')
    parser.add_argument('--max_batches', default=None, type=int, help='stop after this many self-play batches')
    parser.add_argument('--net_precision', default=None, choices=['strict', 'mx6', 'fp16x2', 'fp16', 'bf16', 'bf16x2', 'fp32'], type=str,
                        help='net engine (policy_value_network.PRECISIONS): strict (default) = measured within 5e-4 absolute of the '
                             'fp32 graph on 64 positions with the LIVE weights after every weight change, falling over mx6 -> fp16x2 -> '
                             'fp32 otherwise (mx6: ~4e-5 of the largest logit; fp16x2: ~6e-6); fp16 = 2x faster, ~1e-3 of the largest logit')
    args = parser.parse_args()

    if int(os.environ.get("WORLD_SIZE", "1")) > 1:   # launched by torch.distributed.run: one rank per GPU
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if os.environ.get("CCHESS_ALL_ON_DEVICE0"):   # tests: the N>1 code path on a one-GPU box (with CCHESS_DIST_BACKEND=gloo)
            local_rank = 0
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        from cchess_zero_amd import parallel as _pl
        cpus = _pl.pin_rank_to_cpus(int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("LOCAL_WORLD_SIZE", os.environ["WORLD_SIZE"])))   # own CPUs per rank
        torch.set_num_threads(max(1, min(4, len(cpus) if cpus else 2)))   # the intra-op pools were sized for the whole box
        if os.environ.get("CCHESS_DIST_BACKEND", "nccl") == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if args.net_precision:
        os.environ["CCHESS_NET_PRECISION"] = args.net_precision

    if args.mode == 'train':
        train_main = cchess_main(args.train_playout, args.batch_size, True, args.search_threads, args.processor, args.num_gpus,
                                 args.res_block_nums, args.human_color, games=args.games)
        train_main.run(args.max_batches)
        if os.environ.get("CCHESS_WEIGHT_DIGEST_DIR"):   # tests: every rank leaves a digest of its replica (they must be equal)
            import hashlib
            import torch
            h = hashlib.sha256()
            for k_, v_ in sorted(train_main.policy_value_netowrk.module.state_dict().items()):
                h.update(k_.encode())
                h.update(v_.detach().cpu().numpy().tobytes())
            with open(os.path.join(os.environ["CCHESS_WEIGHT_DIGEST_DIR"], "rank%s.txt" % os.environ.get("RANK", "0")), "w") as f:
                f.write("%s %d %d\n" % (h.hexdigest(), train_main.global_step, len(train_main.data_buffer)))
    elif args.mode == 'play':
        _play_headless(args)
