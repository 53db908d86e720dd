"""Shared driver: run a lock-step engine (C oracle or HIP) through a golden mcts.json case.

`engine` needs: reset(boards, side, rr), select(mode)->(planes f32 [G,9,10,14], need u8 [G]),
expand_backup(logits, value), root_stats()->dict, advance(labels), tree_dump(g), status().
"""
import hashlib

import numpy as np

import fakenet

PIECES = ".KARBNPCkarbnpc"


def fen_to_board(fen):
    b = np.zeros(90, np.uint8)
    y = 0
    for row in fen.split("/"):
        x = 0
        for ch in row:
            if ch.isdigit():
                x += int(ch)
            else:
                b[y * 9 + x] = PIECES.index(ch)
                x += 1
        y += 1
    return b


def tree_digest(rec):
    return hashlib.sha256(np.ascontiguousarray(rec, dtype=np.int32).tobytes()).hexdigest()


def run_cases(engine, cases):
    """Runs all `cases` concurrently (one tree per case, lock-step) ply by ply.
    Returns per case: list of per-ply dict(root=[(label,N,Wbits,Qbits,Pbits)], digest, records, evals)."""
    G = len(cases)
    logs = [[] for _ in range(G)]
    fwds = [fakenet.make_forward(c["mode"], c["salt"], logs[i]) for i, c in enumerate(cases)]
    boards = np.stack([fen_to_board(c["plies"][0]["state"]) for c in cases])
    side = np.array([1 if c["plies"][0]["player"] == "b" else 0 for c in cases], np.uint8)
    rr = np.array([c["plies"][0]["rr"] for c in cases], np.int32)
    engine.reset(boards, side, rr)
    results = [[] for _ in range(G)]
    max_plies = max(len(c["plies"]) for c in cases)
    for ply in range(max_plies):
        active = np.array([ply < len(c["plies"]) for c in cases])
        playouts = np.array([c["plies"][ply]["playouts"] if ply < len(c["plies"]) else 0 for c in cases])
        done = np.zeros(G, np.int64)
        step = 0
        while True:
            mode = 0 if step == 0 else 1
            if mode == 1 and not np.any(active & (done < playouts)):
                break
            planes, need = engine.select(mode, mask=(active & ((done < playouts) | (mode == 0))))
            logits = np.zeros((G, 2086), np.float32)
            value = np.zeros((G, 1), np.float32)
            for g in range(G):
                if need[g]:
                    lg, v = fwds[g](planes[g])
                    logits[g] = lg[0]
                    value[g] = v[0]
            engine.expand_backup(logits, value)
            if mode == 1:
                done += (active & (done < playouts)).astype(np.int64)
            step += 1
        st = engine.root_stats()
        played = np.full(G, 0xFFFF, np.uint16)
        for g in range(G):
            if not active[g]:
                continue
            n = int(st["count"][g])
            root = [(int(st["label"][g, i]), int(st["N"][g, i]), int(st["W"][g, i].view(np.uint32)),
                     int(st["Q"][g, i].view(np.uint32)), int(st["P"][g, i].view(np.uint32))) for i in range(n)]
            rec = engine.tree_dump(g)
            visited = rec[rec[:, 2] > 0]
            results[g].append(dict(root=root, digest=tree_digest(rec), records=len(rec), evals=len(logs[g]),
                                   max_level=int(visited[:, 0].max()) if len(visited) else -1))
            played[g] = cases[g]["plies"][ply]["played"]
            if ply + 1 >= len(cases[g]["plies"]):
                played[g] = 0xFFFF
        engine.advance(played)
    return results, logs


def check_against_golden(results, logs, cases):
    for g, c in enumerate(cases):
        for ply, gp in enumerate(c["plies"]):
            r = results[g][ply]
            exp_root = [tuple(x) for x in gp["root"]]
            assert r["root"] == exp_root, "case %s ply %d: root children differ" % (c["name"], ply)
            assert r["records"] == gp["tree_records"], "case %s ply %d: tree size" % (c["name"], ply)
            assert r["digest"] == gp["tree_sha256"], "case %s ply %d: whole-tree digest" % (c["name"], ply)
            assert r["evals"] == gp["evals"], "case %s ply %d: number of net evaluations" % (c["name"], ply)
            if "max_level" in gp:
                assert r["max_level"] == gp["max_level"], "case %s ply %d: deepest visited level" % (c["name"], ply)
        if "eval_keys" in c:
            assert ["%016x" % k for k in logs[g]] == c["eval_keys"], "case %s: evaluated positions / order" % c["name"]
        else:   # mcts_deep.json stores a digest of the (long) key list
            assert hashlib.sha256(",".join("%016x" % k for k in logs[g]).encode()).hexdigest() == c["eval_keys_sha256"], \
                "case %s: evaluated positions / order" % c["name"]


def fc_logits_restated(z, w, b):
    """The arithmetic k_expand_backup<FC> documents (cz_search.hip), restated in NumPy float32 for ALL labels:
    products p[k] = w[k] * x[k] (x = the (h,w,c) flatten of the two policy channels, padded to 192 with zeros);
    16 partial sums of 12 consecutive products each, added in order; four symmetric folding steps; + bias.
    z [G,90,3], w [2086,180], b [2086] -> [G,2086] float32."""
    G = z.shape[0]
    x = np.zeros((G, 192), np.float32)
    x[:, :180] = z[:, :, :2].reshape(G, 180)
    wp = np.zeros((2086, 192), np.float32)
    wp[:, :180] = w
    p = (wp[None, :, :] * x[:, None, :]).astype(np.float32).reshape(G, 2086, 16, 12)
    s = p[..., 0].copy()
    for k in range(1, 12):
        s = (s + p[..., k]).astype(np.float32)
    l = np.arange(16)
    for partner in (15 - l, (l & 8) | (7 - (l & 7)), (l & 12) | (3 - (l & 3)), l ^ 1):
        s = (s + s[..., partner]).astype(np.float32)
    return (s[..., 0] + b[None, :].astype(np.float32)).astype(np.float32)
