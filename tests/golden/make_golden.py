#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ by IMPORTING the unmodified
Python reference from /root/reference (build container only; the reference never
travels to the GPU box -- only the data files this script writes do).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What it captures:
  items.json            ItemsGenerator.items_generator(seed) lists
  game_rules.npz        getValidMoves / getNextState / has_valid_moves over random rollouts,
                        random (unreachable) boards and hand-built edge cases
  ranked_reward.json    getRankedReward known answers
  q_update.json         the backup expression (N*Q+v)/(N+1) evaluated by NumPy itself for every type mix
  mcts_*.npz            MCTS.getActionProb root counts per move + full tree dumps under table evaluators
  nnet_*.npz            NNetWrapper.predict outputs for seeded weights and for a bundled trained checkpoint
  train_c2.npz          NNetWrapper.train: weights before / after a few Adam steps, loss_pi / loss_v on a fixed batch

Only stand-ins for absent, unused third-party imports are injected (torchvision, wandb);
RNG draws inside the reference (np.random.choice in the tie branch) are made deterministic by
patching numpy's function object for the duration of a call -- the reference source is untouched.
"""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
REF = "/root/reference/xw_mcts"
sys.path.insert(0, REF)
sys.dont_write_bytecode = True
for name in ("torchvision", "wandb"):
    if name not in sys.modules:
        mod = types.ModuleType(name)
        mod.datasets = types.ModuleType(name + ".datasets")
        mod.transforms = types.ModuleType(name + ".transforms")
        mod.log = lambda *a, **k: None
        sys.modules[name] = mod
        sys.modules[name + ".datasets"] = mod.datasets
        sys.modules[name + ".transforms"] = mod.transforms

import evaluators as ev  # noqa: E402
from binpacking.BinPackingGame import BinPackingGame, ItemsGenerator  # noqa: E402
from MCTS_bpp import MCTS  # noqa: E402

META = {"numpy": np.__version__, "generator": "tests/golden/make_golden.py", "reference": "Wang-Xiaoyang/resource_packing_self_play@v1"}


class Args(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def jdump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, separators=(",", ":"))
    print("wrote", name)


# ---------------------------------------------------------------------------------------------
def gen_items():
    out = {"meta": META, "cases": []}
    for (bw, bh, n), seeds in [((10, 10, 8), range(0, 24)), ((20, 20, 32), range(100, 116)), ((15, 15, 10), range(0, 16)),
                               ((50, 50, 128), range(100, 104)), ((15, 7, 10), range(0, 6)), ((15, 2, 10), range(0, 4)),
                               ((20, 11, 32), range(0, 4))]:
        gen = ItemsGenerator(bw, bh, n)
        for s in seeds:
            out["cases"].append({"bin_w": bw, "bin_h": bh, "n": n, "seed": int(s),
                                 "items": [[int(v) for v in it] for it in gen.items_generator(int(s))]})
    jdump("items.json", out)


def items_for(bw, bh, n, seed, gen_h=None):
    gen = ItemsGenerator(bw, gen_h or bh, n)
    return [[int(v) for v in it] for it in gen.items_generator(seed)]


def valid_mask(g, state):
    try:
        return np.asarray(g.getValidMoves(state), dtype=np.uint8)
    except AssertionError:  # BinPackingGame.py:89: no legal move
        return np.zeros(g.getActionSize(), dtype=np.uint8)


def gen_game_rules():
    rng = np.random.default_rng(20201108)
    recs = {k: [] for k in ("W", "H", "N", "rows", "rem", "iw", "ih", "valid", "has", "action", "next_rows", "next_rem", "kind")}

    def record(g, state, action, kind):
        rows, rem, iw, ih = ev.pack_state(state)
        recs["W"].append(g.bin_width); recs["H"].append(g.bin_height); recs["N"].append(g.num_items)
        recs["rows"].append(rows); recs["rem"].append(rem); recs["iw"].append(iw); recs["ih"].append(ih)
        recs["valid"].append(valid_mask(g, state)); recs["has"].append(bool(g.has_valid_moves(state)))
        recs["kind"].append(kind)
        if action is None:
            recs["action"].append(-1); recs["next_rows"].append(rows); recs["next_rem"].append(rem)
            return None
        b, it = g.getNextState(state[0], int(action), state[1:])
        nxt = g.getBinItem(b, it)
        nrows, nrem, _, _ = ev.pack_state(nxt)
        recs["action"].append(int(action)); recs["next_rows"].append(nrows); recs["next_rem"].append(nrem)
        return nxt

    # (1) random rollouts over legal moves, several configs
    for (w, h, n, gen_h), seeds in [((10, 10, 8, None), range(12)), ((20, 20, 32, None), range(100, 106)),
                                    ((15, 15, 10, None), range(6)), ((15, 15, 10, 7), range(3)),
                                    ((50, 50, 128, None), range(100, 101)), ((33, 40, 20, None), range(2))]:
        for seed in seeds:
            g = BinPackingGame(w, h, n, 1)
            items = items_for(w, h, n, seed, gen_h)
            state = g.getBinItem(g.getInitBoard(), g.getInitItems(items))
            while True:
                v = valid_mask(g, state)
                if v.sum() == 0:
                    record(g, state, None, 0)
                    break
                a = int(rng.choice(np.nonzero(v)[0]))
                state = record(g, state, a, 0)
    # (2) random 0/1 boards (mostly unreachable) with random item sets and arbitrary placements,
    #     including x + w > W, which NumPy slicing clips (BinPackingLogic.py:104-105)
    for (w, h, n) in [(10, 10, 8), (20, 20, 32), (7, 9, 5), (32, 12, 9), (33, 6, 4), (64, 64, 3)]:
        for t in range(14 if w < 64 else 4):
            g = BinPackingGame(w, h, n, 1)
            dens = rng.choice([0.05, 0.2, 0.5, 0.8, 0.95])
            board = (rng.random((h, w)) < dens).astype(np.int64)
            if t % 3 == 0:  # staircase-like boards: columns filled from the bottom row 0 upwards
                heights = rng.integers(0, h + 1, size=w)
                board = (np.arange(h)[:, None] < heights[None, :]).astype(np.int64)
            items = [[int(rng.integers(1, w + 1)), int(rng.integers(1, h + 1)), 0, 0] for _ in range(n)]
            planes = g.getInitItems(items)
            for i in range(n):
                if rng.random() < 0.3:
                    planes[i] = planes[i] * 0
            if all(p.sum() == 0 for p in planes):
                planes = g.getInitItems(items)
            state = g.getBinItem(board, planes)
            live = [i for i in range(n) if planes[i].sum() > 0]
            i = int(rng.choice(live)); x = int(rng.integers(0, w))
            record(g, state, i * w + x, 1)
    # (3) hand-built edge cases
    def hand(w, h, items, cells, placed, action):
        g = BinPackingGame(w, h, len(items), 1)
        board = np.zeros((h, w), dtype=np.int64)
        for (r, c) in cells:
            board[r, c] = 1
        planes = g.getInitItems(items)
        for i in placed:
            planes[i] = planes[i] * 0
        record(g, g.getBinItem(board, planes), action, 2)
    # partial placement: 2x2 item, columns 0-1 have only one free row -> only 2 cells get filled
    hand(4, 3, [[2, 2, 0, 0], [1, 1, 0, 0]], [(0, 0), (0, 1), (1, 0), (1, 1)], [], 0)
    # adjacency fall-through t = H-1: no empty window row at j=1
    hand(4, 3, [[2, 1, 0, 0], [1, 1, 0, 0]], [(0, 1), (1, 2), (2, 1), (2, 0)], [], 1)
    hand(4, 3, [[2, 1, 0, 0], [1, 1, 0, 0]], [(0, 1), (1, 2), (2, 1)], [], 1)
    # full column / last item / full board
    hand(3, 3, [[1, 3, 0, 0], [1, 1, 0, 0]], [(0, 0), (1, 0), (2, 0)], [], 1)
    hand(3, 3, [[1, 3, 0, 0], [2, 3, 0, 0]], [(0, 0), (1, 0), (2, 0)], [0], 3 + 1)
    hand(2, 2, [[1, 1, 0, 0]], [(0, 0), (0, 1), (1, 0), (1, 1)], [], 0)
    # non-contiguous fill: free rows 0 and 2 under the window
    hand(5, 4, [[2, 2, 0, 0], [1, 1, 0, 0]], [(0, 0), (1, 1), (1, 2), (2, 0)], [], 1)
    # empty board, widest item
    hand(6, 5, [[6, 5, 0, 0], [1, 1, 0, 0]], [], [], 0)

    nrec = len(recs["W"])
    maxh = max(recs["H"]); maxn = max(recs["N"]); maxa = max(w * n for w, n in zip(recs["W"], recs["N"]))
    out = {
        "W": np.array(recs["W"], np.int32), "H": np.array(recs["H"], np.int32), "N": np.array(recs["N"], np.int32),
        "kind": np.array(recs["kind"], np.uint8), "has": np.array(recs["has"], np.uint8),
        "action": np.array(recs["action"], np.int32),
        "rows": np.zeros((nrec, maxh), np.uint64), "next_rows": np.zeros((nrec, maxh), np.uint64),
        "rem": np.zeros((nrec, maxn), np.uint8), "next_rem": np.zeros((nrec, maxn), np.uint8),
        "iw": np.zeros((nrec, maxn), np.uint8), "ih": np.zeros((nrec, maxn), np.uint8),
        "valid_bits": np.zeros((nrec, (maxa + 7) // 8), np.uint8),
    }
    for i in range(nrec):
        H, N, A = recs["H"][i], recs["N"][i], recs["W"][i] * recs["N"][i]
        out["rows"][i, :H] = recs["rows"][i]; out["next_rows"][i, :H] = recs["next_rows"][i]
        out["rem"][i, :N] = recs["rem"][i]; out["next_rem"][i, :N] = recs["next_rem"][i]
        out["iw"][i, :N] = recs["iw"][i]; out["ih"][i, :N] = recs["ih"][i]
        bits = np.packbits(recs["valid"][i], bitorder="little")
        out["valid_bits"][i, :len(bits)] = bits
    np.savez_compressed(os.path.join(HERE, "game_rules.npz"), meta=json.dumps(META), **out)
    print("wrote game_rules.npz", nrec, "records")


# ---------------------------------------------------------------------------------------------
class _TieChoice:
    """Context manager: np.random.choice returns np.int64(value) (or a sentinel) while active."""
    def __init__(self, value):
        self.value = value
    def __enter__(self):
        self.orig = np.random.choice
        np.random.choice = lambda *a, **k: np.int64(self.value)
        return self
    def __exit__(self, *exc):
        np.random.choice = self.orig


def gen_ranked_reward():
    rng = np.random.default_rng(7)
    cases = []
    def add(w, h, n, items, cells_board, area, buf, alpha):
        g = BinPackingGame(w, h, n, 1)
        planes = g.getInitItems(items)  # sets max_h
        planes = [p * 0 for p in planes]
        state = g.getBinItem(cells_board, planes)
        with _TieChoice(2):
            ranked, r = g.getRankedReward(state, area, list(buf), alpha)
        rows = ev.pack_board(cells_board)
        cases.append({"W": w, "H": h, "max_h": int(g.max_h), "rows": [int(x) for x in rows], "area": int(area),
                      "buf": [float(x) for x in buf], "alpha": alpha, "ranked": int(ranked), "r": float(r)})
    for (w, h) in [(10, 10), (20, 20), (15, 15), (5, 8)]:
        for t in range(10):
            top = int(rng.integers(1, h + 1))
            board = np.zeros((h, w), dtype=np.int64)
            board[:top, :] = 1
            if t % 4 == 1:  # ragged top row
                board[top - 1, int(rng.integers(0, w)):] = 0
                if board[top - 1].sum() == 0:
                    board[top - 1, 0] = 1
            area = int(board.sum())
            items = [[int(rng.integers(1, w + 1)), int(rng.integers(1, max(2, top))), 0, 0] for _ in range(4)]
            r_exact = None
            bufs = [[], [0.9], [0.5, 0.9, 0.7], [0.8, 1.0, 0.6, 0.9], list(np.round(rng.uniform(0.8, 1.0, 100), 6))]
            # buffers made of plausible reward ratios so that r == bl ties occur
            ratios = sorted({a / b for a in range(1, h + 1) for b in range(a, h + 1)})
            bufs.append([float(x) for x in rng.choice(ratios, size=40)])
            mh = max(it[1] for it in items)
            r_here = max(np.ceil(area / w), mh) / top
            bufs.append([float(r_here)] * 5)
            for buf in bufs:
                for alpha in (0.75, 0.5, 0.1):
                    add(w, h, 4, items, board, area, buf, alpha)
            # area mismatch -> r = 0
            add(w, h, 4, items, board, area + 1, [0.9, 0.8], 0.75)
            add(w, h, 4, items, board, area + 1, [], 0.75)
            add(w, h, 4, items, board, area + 1, [0.0, 0.0], 0.75)
    # empty board (top row = 1 by the loop fall-through)
    add(6, 6, 2, [[1, 1, 0, 0], [2, 2, 0, 0]], np.zeros((6, 6), dtype=np.int64), 0, [0.5], 0.75)
    jdump("ranked_reward.json", {"meta": META, "tie_sentinel": 2, "cases": cases})


# ---------------------------------------------------------------------------------------------
def gen_q_update():
    """(Nsa*Qsa + v)/(Nsa+1) (MCTS_bpp.py:131) evaluated by NumPy/CPython for chains of mixed v types."""
    rng = np.random.default_rng(11)
    chains = []
    def mk(kind):
        if kind == 0:
            return int(rng.choice([1, -1]))
        if kind == 1:
            return np.array([np.float32(rng.uniform(-1, 1))], dtype=np.float32)
        return np.int64(rng.choice([1, -1]))
    patterns = [[0] * 12, [1] * 12, [2] * 6, [0, 0, 0, 0, 0, 1, 1, 0, 1], [0, 0, 0, 2, 1, 0], [1, 1, 2, 1, 0, 1], [0, 1, 2, 0, 1, 2],
                [2, 0, 0, 1], [0, 0, 0, 0, 0, 0, 0, 1, 1, 1]]
    for rep in range(40):
        for pat in patterns:
            if rep >= 4:
                pat = [int(k) for k in rng.choice([0, 0, 1, 1, 2], size=int(rng.integers(3, 14)))]
            steps = []
            Q = None
            N = 0
            for kind in pat:
                v = mk(kind)
                if N == 0:
                    Q = v  # :135
                else:
                    Q = (N * Q + v) / (N + 1)  # :131
                N += 1
                if isinstance(Q, np.ndarray):
                    qk = 1 if Q.dtype == np.float32 else 2
                    qv = float(Q[0])
                elif isinstance(Q, (np.floating, np.integer)):
                    qk, qv = 2, float(Q)
                else:
                    qk, qv = 0, float(Q)
                steps.append({"v": float(np.asarray(v).reshape(-1)[0]), "v_kind": kind, "q": qv.hex(), "q_kind": qk})
            chains.append(steps)
    jdump("q_update.json", {"meta": META, "kinds": {"0": "python int/float (weak)", "1": "float32 array", "2": "int64/float64 (strong)"},
                            "chains": chains})


# ---------------------------------------------------------------------------------------------
class TableNet:
    def __init__(self, game, kind, salt):
        self.game, self.kind, self.salt = game, kind, salt
        self.calls = 0
    def predict(self, state):
        self.calls += 1
        rows, rem, _, _ = ev.pack_state(state)
        return ev.table_eval(self.kind, rows, rem, self.game.getActionSize(), self.salt)


class TieGame(BinPackingGame):
    """Reference game whose tie branch draws ev.tie_value(state) instead of OS-seeded randomness."""
    tie_salt = 0
    def getRankedReward(self, total_board, items_total_area, rewards_list, alpha):
        rows, rem, _, _ = ev.pack_state(total_board)
        with _TieChoice(ev.tie_value(rows, rem, self.tie_salt)):
            return BinPackingGame.getRankedReward(self, total_board, items_total_area, rewards_list, alpha)


def q_kind_of(q):
    if isinstance(q, np.ndarray):
        return (1 if q.dtype == np.float32 else 2), float(q.reshape(-1)[0])
    if isinstance(q, (np.floating, np.integer)):
        return 2, float(q)
    return 0, float(q)


def dump_tree(g, mcts):
    """All of Es / Ps / Ns / Vs / Nsa / Qsa keyed by the packed state."""
    H, W, N = g.bin_height, g.bin_width, g.num_items
    plane = H * W * 8
    def unkey(s):
        arr = np.frombuffer(s, dtype=np.int64).reshape(N + 1, H, W)
        rows, rem, _, _ = ev.pack_state(arr)
        return rows, rem
    keys = list(mcts.Es.keys())
    index = {s: i for i, s in enumerate(keys)}
    node_rows = np.zeros((len(keys), H), np.uint64); node_rem = np.zeros((len(keys), N), np.uint8)
    node_es = np.zeros(len(keys), np.int8); node_es_kind = np.zeros(len(keys), np.uint8)
    node_exp = np.zeros(len(keys), np.uint8); node_ns = np.zeros(len(keys), np.uint32)
    e_node, e_act, e_p, e_n, e_q, e_qk = [], [], [], [], [], []
    for s, i in index.items():
        node_rows[i], node_rem[i] = unkey(s)
        es = mcts.Es[s]
        node_es[i] = int(es)
        node_es_kind[i] = 2 if isinstance(es, np.integer) else 0
        if s in mcts.Ps:
            node_exp[i] = 1
            node_ns[i] = mcts.Ns[s]
            P = mcts.Ps[s]; V = mcts.Vs[s]
            assert P.dtype == np.float64
            for a in np.nonzero(V)[0]:
                a = int(a)
                e_node.append(i); e_act.append(a); e_p.append(float(P[a]))
                if (s, a) in mcts.Nsa:
                    k, q = q_kind_of(mcts.Qsa[(s, a)])
                    e_n.append(int(mcts.Nsa[(s, a)])); e_q.append(q); e_qk.append(k)
                else:
                    e_n.append(0); e_q.append(0.0); e_qk.append(0)
            # invalid entries of P must be exactly zero
            assert float(np.abs(P[np.asarray(V) == 0]).sum()) == 0.0
    return dict(node_rows=node_rows, node_rem=node_rem, node_es=node_es, node_es_kind=node_es_kind, node_exp=node_exp,
                node_ns=node_ns, e_node=np.array(e_node, np.int32), e_act=np.array(e_act, np.int32),
                e_p=np.array(e_p, np.float64), e_n=np.array(e_n, np.uint32), e_q=np.array(e_q, np.float64),
                e_qk=np.array(e_qk, np.uint8))


def run_mcts_episode(w, h, n, item_seed, sims, kind, salt, buf, alpha=0.75, cpuct=1, gen_h=None, area=None):
    g = TieGame(w, h, n, 1)
    g.tie_salt = salt
    items = items_for(w, h, n, item_seed, gen_h)
    net = TableNet(g, kind, salt)
    args = Args(numMCTSSims=sims, cpuct=cpuct, alpha=alpha)
    mcts = MCTS(g, net, args)
    total_area = area if area is not None else w * (gen_h or h)
    board = g.getInitBoard(); planes = g.getInitItems(items)
    counts_per_move, actions = [], []
    outcome, score = 0, 0.0
    while True:
        state = g.getBinItem(board, planes)
        pi = mcts.getActionProb(state, total_area, list(buf))
        s = g.stringRepresentation(state)
        counts = np.array([mcts.Nsa.get((s, a), 0) for a in range(g.getActionSize())], dtype=np.uint32)
        assert np.allclose(np.array(pi), counts / counts.sum())
        counts_per_move.append(counts)
        a = int(np.argmax(counts))  # lowest-index argmax (deterministic stand-in for CoachBPP.py:86-87)
        actions.append(a)
        board, planes = g.getNextState(board, a, planes)
        r, sc = g.getGameEnded(g.getBinItem(board, planes), total_area, list(buf), alpha)
        if r != 0:
            outcome, score = int(r), float(sc)
            break
    tree = dump_tree(g, mcts)
    iw = np.array([it[0] for it in items], np.uint8); ih = np.array([it[1] for it in items], np.uint8)
    return dict(W=w, H=h, N=n, sims=sims, kind=kind, salt=salt, alpha=alpha, cpuct=float(cpuct), total_area=total_area,
                buf=np.array(buf, np.float64), item_w=iw, item_h=ih, counts=np.stack(counts_per_move),
                actions=np.array(actions, np.int32), outcome=outcome, score=score, evals=net.calls, **tree)


def gen_mcts():
    rng = np.random.default_rng(3)
    buf100 = [float(x) for x in np.round(rng.uniform(0.8, 1.0, 100), 6)]
    tie_buf = [10 / 11] * 30 + [10 / 12] * 30 + [1.0] * 10 + [10 / 13] * 30  # many equal scores -> r == bl ties
    low_buf = [0.0] * 20  # r == bl == 0 ties when items are discarded
    cases = [
        # (name, w, h, n, item_seed, sims, kind, salt, buf)
        ("c1_uniform", 10, 10, 8, 100, 25, "uniform", 0, []),
        ("c1_hashed_buf", 10, 10, 8, 101, 25, "hashed", 1, buf100),
        ("c2_hashed", 10, 10, 8, 102, 100, "hashed", 2, buf100),
        ("c2_sparse_tie", 10, 10, 8, 103, 100, "sparse", 3, tie_buf),
        ("c2_peaked_low", 10, 10, 8, 104, 100, "peaked", 4, low_buf),
        ("c2_uniform_tie", 10, 10, 8, 105, 60, "uniform", 5, low_buf),
        ("w15_hashed", 15, 15, 10, 7, 200, "hashed", 6, buf100),
        ("w15_h7_sparse", 15, 15, 10, 8, 80, "sparse", 7, tie_buf),
        ("c3_uniform", 20, 20, 32, 100, 30, "uniform", 8, buf100),
        ("c3_peaked", 20, 20, 32, 101, 40, "peaked", 9, buf100),
        ("c3_hashed_tie", 20, 20, 32, 102, 30, "hashed", 10, low_buf),
    ]
    for (name, w, h, n, iseed, sims, kind, salt, buf) in cases:
        gen_h = 7 if name == "w15_h7_sparse" else None
        rec = run_mcts_episode(w, h, n, iseed, sims, kind, salt, buf, gen_h=gen_h)
        np.savez_compressed(os.path.join(HERE, "mcts_%s.npz" % name), meta=json.dumps(META), **rec)
        kinds = np.bincount(rec["e_qk"][rec["e_n"] > 0], minlength=3)
        print("wrote mcts_%s.npz moves=%d nodes=%d edges=%d evals=%d outcome=%d score=%.4f qkinds=%s es_strong=%d" % (
            name, len(rec["actions"]), len(rec["node_es"]), len(rec["e_node"]), rec["evals"], rec["outcome"], rec["score"],
            kinds.tolist(), int((rec["node_es_kind"] == 2).sum())))


# ---------------------------------------------------------------------------------------------
def gen_nnet():
    import torch
    from binpacking.pytorch.NNet import NNetWrapper
    META["torch"] = torch.__version__
    rng = np.random.default_rng(5)
    def states_for(w, h, n, count):
        g = BinPackingGame(w, h, n, 1)
        out = []
        seed = 100
        while len(out) < count:
            items = items_for(w, h, n, seed); seed += 1
            state = g.getBinItem(g.getInitBoard(), g.getInitItems(items))
            while len(out) < count:
                out.append(state)
                v = valid_mask(g, state)
                if v.sum() == 0:
                    break
                b, it = g.getNextState(state[0], int(rng.choice(np.nonzero(v)[0])), state[1:])
                state = g.getBinItem(b, it)
        return g, out
    def run(name, w, h, n, count, state_dict=None, seed=0):
        g, states = states_for(w, h, n, count)
        args = Args(cuda=False, num_items=n, num_bins=1, epochs=1, batch_size=8)
        torch.manual_seed(seed)
        net = NNetWrapper(g, args)
        if state_dict is not None:
            net.nnet.load_state_dict(state_dict)
        pis, vs = [], []
        for s in states:
            pi, v = net.predict(s)
            pis.append(pi); vs.append(v)
        weights = {"w__" + k: t.detach().numpy() for k, t in net.nnet.state_dict().items()}
        rows = np.stack([ev.pack_state(s)[0] for s in states]); rem = np.stack([ev.pack_state(s)[1] for s in states])
        iw = np.stack([np.array(s[1:, 0, :].sum(axis=1)) for s in states]); ih = np.stack([np.array(s[1:, :, 0].sum(axis=1)) for s in states])
        np.savez_compressed(os.path.join(HERE, "nnet_%s.npz" % name), meta=json.dumps(META), W=w, H=h, N=n,
                            planes=np.stack(states).astype(np.uint8), rows=rows, rem=rem, pi=np.stack(pis), v=np.stack(vs), **weights)
        print("wrote nnet_%s.npz" % name, np.stack(pis).shape)
    run("c2_seed0", 10, 10, 8, 16)
    run("c3_seed0", 20, 20, 32, 6)
    ck = os.path.join(REF, "wandb", "run-20201113_144231-15y0rcng", "temp", "temp.pth.tar")
    sd = torch.load(ck, map_location="cpu", weights_only=True)["state_dict"]
    run("w15_trained", 15, 15, 10, 12, state_dict=sd)


def gen_nnet_f64():
    """Float64 ground truth for the nnet_*.npz fixtures: the reference's own BinPackingNNet (BinpackingNNet.py:50-81) with the stored
    weights cast to double, on the stored states.  Shows how far the reference's float32 outputs themselves are from the exact
    forward of the same weights (2e-5 on pi for the trained 15x15 checkpoint) -- the yardstick tests/test_gpu_nnet.py holds the HIP
    evaluator to where 1e-5 against the float32 reference is below that noise floor."""
    import torch
    from binpacking.pytorch.BinpackingNNet import BinPackingNNet as RefNet
    out = {"meta": json.dumps(dict(META, torch=torch.__version__))}
    for name in ("c2_seed0", "c3_seed0", "w15_trained", "c5_seed0"):
        path = os.path.join(HERE, "nnet_%s.npz" % name)
        if not os.path.exists(path):
            continue
        d = np.load(path)
        w, h, n = int(d["W"]), int(d["H"]), int(d["N"])
        g = BinPackingGame(w, h, n, 1)
        torch.manual_seed(int(d["seed"])) if "seed" in d.files else None
        net = RefNet(g, Args(num_items=n, num_bins=1))
        if any(k.startswith("w__") for k in d.files):
            net.load_state_dict({k[3:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("w__")})
        net = net.double().eval()
        with torch.no_grad():
            lp, v = net(torch.from_numpy(d["planes"].astype(np.float64)))
        out[name + "__pi64"] = torch.exp(lp).numpy(); out[name + "__v64"] = v.numpy()
        print("%s: reference f32 vs f64  pi %.3e  v %.3e" % (name, np.abs(d["pi"] - out[name + "__pi64"]).max(), np.abs(d["v"] - out[name + "__v64"]).max()))
    np.savez_compressed(os.path.join(HERE, "nnet_f64.npz"), **out)
    print("wrote nnet_f64.npz")


def gen_nnet_c5():
    """NNetWrapper.predict at BASELINE configs[4] (50x50 bin, 128 items): 3 states.  The 2.16 M weights (8.7 MB) are not stored:
    they are torch.manual_seed(0)'s initialisation, which the package's module reproduces exactly (same layer order); the
    fixture keeps their SHA-256 so a test can tell a changed initialiser from a wrong forward."""
    import hashlib
    import torch
    from binpacking.pytorch.NNet import NNetWrapper
    from resource_packing_self_play_amd.binpacking.pytorch.BinpackingNNet import BinPackingNNet as OwnNet
    w, h, n, seed = 50, 50, 128, 0
    g = BinPackingGame(w, h, n, 1)
    rng = np.random.default_rng(11)
    items = items_for(w, h, n, 100)
    state = g.getBinItem(g.getInitBoard(), g.getInitItems(items))
    states = []
    for step in range(41):
        if step in (0, 12, 40):
            states.append(state)
        v = valid_mask(g, state)
        b, it = g.getNextState(state[0], int(rng.choice(np.nonzero(v)[0])), state[1:])
        state = g.getBinItem(b, it)
    args = Args(cuda=False, num_items=n, num_bins=1, epochs=1, batch_size=8)
    torch.manual_seed(seed)
    net = NNetWrapper(g, args)
    torch.manual_seed(seed)
    own = OwnNet(g, args)
    sd, so = net.nnet.state_dict(), own.state_dict()
    assert list(sd) == list(so) and all(torch.equal(sd[k], so[k]) for k in sd), "the package's module no longer initialises like the reference's"
    digest = hashlib.sha256(b"".join(sd[k].numpy().tobytes() for k in sd)).hexdigest()
    pis, vs = [], []
    for s in states:
        pi, v = net.predict(s)
        pis.append(pi); vs.append(v)
    rows = np.stack([ev.pack_state(s)[0] for s in states]); rem = np.stack([ev.pack_state(s)[1] for s in states])
    np.savez_compressed(os.path.join(HERE, "nnet_c5_seed0.npz"), meta=json.dumps(dict(META, torch=torch.__version__)), W=w, H=h, N=n, seed=seed,
                        weights_sha256=digest, planes=np.stack(states).astype(np.uint8), rows=rows, rem=rem, pi=np.stack(pis), v=np.stack(vs))
    print("wrote nnet_c5_seed0.npz", np.stack(pis).shape, digest[:16])


def gen_coach():
    """Two iterations of the reference's CoachBPP.learn (CoachBPP.py:101-196) -- the second one greedy (iterStepThreshold = 1, :132) --
    with everything random pinned: np.random.seed() without an argument (the OS-entropy reseeds at :86 and :117) is ignored after
    one np.random.seed(4242), np.random.choice picks the lowest index of the largest probability (:87; MCTS_bpp.py:46 picks the
    lowest best action), the `r == bl` tie draws ev.tie_value(state) (TieGame), the evaluator is the hashed table evaluator, and
    nnet.train / save_checkpoint / wandb.log are recorded no-ops.  Stored per episode: generator seed, generator height, the R2
    buffer as it stood BEFORE the episode (the reference appends after every episode, :134), the (state, pi, r) tuples (:99) and
    the score; per iteration: the buffer after the trim (:136-139) and the logged metrics (:143-147)."""
    import tempfile
    import wandb
    from CoachBPP import CoachBPP
    w, h, n, sims, salt = 10, 10, 8, 20, 17

    class CoachNet:
        def __init__(self, game, args):
            self.game, self.kind, self.salt, self.trained = game, args.table_kind, args.table_salt, []
        def predict(self, state):
            rows, rem, _, _ = ev.pack_state(state)
            return ev.table_eval(self.kind, rows, rem, self.game.getActionSize(), self.salt)
        def train(self, examples):
            self.trained.append(len(examples))
        def save_checkpoint(self, folder, filename):
            pass

    g = TieGame(w, h, n, 1)
    g.tie_salt = salt
    tmp = tempfile.mkdtemp()
    args = Args(numIters=2, numEps=8, iterStepThreshold=1, maxlenOfQueue=200000, numMCTSSims=sims, cpuct=1, alpha=0.75, seed=100,
                numItersForTrainExamplesHistory=50, numScoresForRank=14, binH_min=6, binH=10, numItems=n, checkpoint=tmp,
                table_kind="hashed", table_salt=salt)
    gen = ItemsGenerator(w, h, n)
    # distinct values around the 75 % quantile: the threshold bl moves when an append leaves floor(len * alpha) unchanged
    initial = [0.5, 0.6, 0.7, 0.75, 0.78, 0.8, 0.8125, 0.875, 0.9, 1.0]
    coach = CoachBPP(g, CoachNet(g, args), gen.items_generator(args.seed), w * h, gen, args, saved_rewards_list=list(initial))
    episodes, seeds, logged = [], [], []
    orig_exec, orig_gen = coach.executeEpisode, gen.items_generator
    def rec_exec(greedy=False):
        before = [float(x) for x in coach.rewards_list]
        ex = orig_exec(greedy)
        episodes.append(dict(greedy=bool(greedy), seed=seeds[-1], bin_height=int(gen.bin_height), total_area=int(coach.items_total_area),
                             items=np.array(coach.items_list)[:, :2].astype(np.uint8), before=before, score=float(coach.ep_score), examples=ex))
        return ex
    def rec_gen(seed):
        seeds.append(int(seed))
        return orig_gen(seed)
    coach.executeEpisode, gen.items_generator = rec_exec, rec_gen
    orig_seed, orig_choice, orig_log = np.random.seed, np.random.choice, getattr(wandb, "log", None)
    def fake_seed(seed=None):
        if seed is not None:
            orig_seed(seed)
    def fake_choice(a, size=None, replace=True, p=None):
        idx = int(np.argmax(np.asarray(p))) if p is not None else 0
        return idx if isinstance(a, (int, np.integer)) else np.asarray(a).reshape(-1)[idx]
    wandb.log = lambda d, step=None: logged.append((int(step), {k: float(v) for k, v in d.items()}))
    np.random.seed(4242)
    np.random.seed, np.random.choice = fake_seed, fake_choice
    try:
        coach.learn()
    finally:
        np.random.seed, np.random.choice = orig_seed, orig_choice
        if orig_log is not None:
            wandb.log = orig_log
    E = args.numEps
    assert len(episodes) == 2 * E and coach.nnet.trained and len(coach.rewards_list) <= args.numScoresForRank
    ex_ep, ex_rows, ex_rem, ex_pi, ex_r = [], [], [], [], []
    for k, e in enumerate(episodes):
        for state, pi, r in e["examples"]:
            rows, rem, _, _ = ev.pack_state(state)
            ex_ep.append(k); ex_rows.append(rows); ex_rem.append(rem); ex_pi.append(np.asarray(pi, np.float64)); ex_r.append(int(r))
    blen = max(len(e["before"]) for e in episodes)
    before = np.full((len(episodes), blen), np.nan)
    for k, e in enumerate(episodes):
        before[k, :len(e["before"])] = e["before"]
    metrics = {}
    for step, dd in logged:
        metrics.setdefault(step, {}).update(dd)
    np.savez_compressed(os.path.join(HERE, "coach_c1.npz"), meta=json.dumps(META), W=w, H=h, N=n, sims=sims, salt=salt, kind="hashed", alpha=args.alpha,
                        numEps=E, numIters=2, iterStepThreshold=1, numScoresForRank=args.numScoresForRank, binH_min=args.binH_min, binH=args.binH,
                        initial=np.array(initial), ep_seed=np.array([e["seed"] for e in episodes], np.int64),
                        ep_bin_height=np.array([e["bin_height"] for e in episodes], np.int32), ep_area=np.array([e["total_area"] for e in episodes], np.int32),
                        ep_items=np.stack([e["items"] for e in episodes]), ep_greedy=np.array([e["greedy"] for e in episodes]),
                        ep_score=np.array([e["score"] for e in episodes]), ep_before=before, ep_before_len=np.array([len(e["before"]) for e in episodes], np.int32),
                        after_iter1=np.array(episodes[E]["before"]), after_iter2=np.array([float(x) for x in coach.rewards_list]),
                        ex_ep=np.array(ex_ep, np.int32), ex_rows=np.stack(ex_rows), ex_rem=np.stack(ex_rem), ex_pi=np.stack(ex_pi), ex_r=np.array(ex_r, np.int8),
                        trained_on=np.array(coach.nnet.trained, np.int64), metrics=json.dumps(metrics))
    def bl_of(buf):
        sb = np.sort(buf)
        return float(sb[int(np.floor(len(sb) * args.alpha)) - 1])
    for it in range(2):
        eps = episodes[it * E:(it + 1) * E]
        print("iteration %d: bin_height %d, scores %s" % (it + 1, eps[0]["bin_height"], [round(e["score"], 4) for e in eps]))
        print("   bl before each episode: %s  (snapshot bl %.4f)" % ([round(bl_of(e["before"]), 4) for e in eps], bl_of(eps[0]["before"])))
        print("   r per episode: %s" % [e["examples"][0][2] for e in eps])
    print("wrote coach_c1.npz: %d examples, trained on %s, metrics %s" % (len(ex_ep), coach.nnet.trained, metrics))


def gen_train_grads():
    """Gradients of the reference's loss (NNet.py:87-91: loss_pi + loss_v) on the fixed batch of train_c2.npz (its first 8 examples)
    at the stored initial weights, through the reference's own module.  Adam turns a gradient that is exactly zero on the CPU but
    1e-9 of rounding noise elsewhere into a step of ~lr, so weights after training are comparable across back ends only where
    the gradient is above noise; the gradients themselves are comparable everywhere."""
    import torch
    from binpacking.pytorch.NNet import NNetWrapper
    d = np.load(os.path.join(HERE, "train_c2.npz"))
    w, h, n = int(d["W"]), int(d["H"]), int(d["N"])
    g = BinPackingGame(w, h, n, 1)
    net = NNetWrapper(g, Args(cuda=False, num_items=n, num_bins=1, epochs=1, batch_size=8))
    net.nnet.load_state_dict({k[3:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("i__")})
    boards = torch.FloatTensor(d["planes"][:8].astype(np.float64))
    tp = torch.FloatTensor(d["pi"][:8]); tv = torch.FloatTensor(d["v"][:8].astype(np.float64))
    net.nnet.train()
    op, ov = net.nnet(boards)
    (net.loss_pi(tp, op) + net.loss_v(tv, ov)).backward()
    grads = {"g__" + k: p.grad.detach().numpy().copy() for k, p in net.nnet.named_parameters()}
    np.savez_compressed(os.path.join(HERE, "train_c2_grads.npz"), meta=json.dumps(dict(META, torch=torch.__version__)), **grads)
    print("wrote train_c2_grads.npz", len(grads), max(float(np.abs(v).max()) for v in grads.values()))


def gen_train():
    """NNetWrapper.train (NNet.py:27-67) for a few steps on CPU from seeded weights and examples: final weights and
    the loss values of loss_pi / loss_v (NNet.py:87-91) on a fixed batch."""
    import io
    import contextlib
    import torch
    from binpacking.pytorch.NNet import NNetWrapper
    META["torch"] = torch.__version__
    w, h, n = 10, 10, 8
    g = BinPackingGame(w, h, n, 1)
    rng = np.random.default_rng(9)
    examples = []
    seed = 300
    while len(examples) < 24:
        items = items_for(w, h, n, seed); seed += 1
        state = g.getBinItem(g.getInitBoard(), g.getInitItems(items))
        while len(examples) < 24:
            v = valid_mask(g, state)
            if v.sum() == 0:
                break
            counts = rng.integers(0, 30, size=v.shape) * v
            if counts.sum() == 0:
                counts = v
            pi = [float(c) / float(counts.sum()) for c in counts]
            examples.append((state, pi, int(rng.choice([1, -1]))))
            b, it = g.getNextState(state[0], int(rng.choice(np.nonzero(v)[0])), state[1:])
            state = g.getBinItem(b, it)
    args = Args(cuda=False, num_items=n, num_bins=1, epochs=2, batch_size=8)
    torch.manual_seed(1)
    net = NNetWrapper(g, args)
    init = {"i__" + k: t.detach().numpy().copy() for k, t in net.nnet.state_dict().items()}
    boards = torch.FloatTensor(np.array([e[0] for e in examples[:8]]).astype(np.float64))
    tp = torch.FloatTensor(np.array([e[1] for e in examples[:8]])); tv = torch.FloatTensor(np.array([e[2] for e in examples[:8]]).astype(np.float64))
    net.nnet.eval()
    with torch.no_grad():
        op, ov = net.nnet(boards)
    l_pi, l_v = float(net.loss_pi(tp, op)), float(net.loss_v(tv, ov))
    np.random.seed(77)
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        net.train(examples)
    final = {"f__" + k: t.detach().numpy().copy() for k, t in net.nnet.state_dict().items()}
    np.savez_compressed(os.path.join(HERE, "train_c2.npz"), meta=json.dumps(META), W=w, H=h, N=n, epochs=2, batch_size=8, np_seed=77,
                        planes=np.stack([e[0] for e in examples]).astype(np.uint8), pi=np.array([e[1] for e in examples], np.float64),
                        v=np.array([e[2] for e in examples], np.int8), loss_pi=l_pi, loss_v=l_v, **init, **final)
    print("wrote train_c2.npz", l_pi, l_v)


if __name__ == "__main__":
    which = sys.argv[1:] or ["items", "rules", "reward", "q", "mcts", "nnet", "nnet_c5", "nnet64", "coach", "train", "train_grads"]
    if "items" in which: gen_items()
    if "rules" in which: gen_game_rules()
    if "reward" in which: gen_ranked_reward()
    if "q" in which: gen_q_update()
    if "mcts" in which: gen_mcts()
    if "nnet" in which: gen_nnet()
    if "nnet_c5" in which: gen_nnet_c5()
    if "nnet64" in which: gen_nnet_f64()
    if "coach" in which: gen_coach()
    if "train" in which: gen_train()
    if "train_grads" in which: gen_train_grads()
