"""Helpers shared by the GPU parity tests (HIP engine through the C ABI vs oracle / golden vectors)."""
import numpy as np

import evaluators as ev


def have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def host_evaluator(kind_of_slot, A, salt_of_slot):
    """-> evaluate(rows, rem, slots) using the table evaluators of tests/evaluators.py."""
    def evaluate(rows, rem, slots):
        pi = np.empty((len(rows), A), np.float32); v = np.empty(len(rows), np.float32)
        for b in range(len(rows)):
            p, val = ev.table_eval(kind_of_slot(int(slots[b])), rows[b], rem[b], A, salt_of_slot(int(slots[b])))
            pi[b] = p; v[b] = val[0]
        return pi, v
    return evaluate


def run_until_idle(eng, evaluate, max_steps=10 ** 7):
    """search_step / evaluate / commit until no slot waits for the evaluator."""
    from resource_packing_self_play_amd import _lib
    steps = 0
    while steps < max_steps:
        n = eng.search_step()
        if n == 0:
            ph = eng.status()[0]
            busy = (_lib.PHASE_RUNNING,) if eng.move_rule == _lib.MOVE_EXTERNAL else (_lib.PHASE_RUNNING, _lib.PHASE_MOVE_READY)
            if not np.isin(ph, busy).any():
                return steps
            continue  # slots between moves (played at the start of the next step) or stopped by the step cap
        rows, rem, slots = eng.leaf_states(n)
        pi, v = evaluate(rows, rem, slots)
        eng.commit_eval_host(pi, v)
        steps += 1
    raise RuntimeError("search did not finish")


def tree_as_dict(d):
    """engine dump -> {(rows bytes, rem bytes): record} like oracle_lib.OracleMCTS.dump()."""
    out = {}
    for i in range(len(d["node_term"])):
        rec = dict(es=int(d["node_term"][i]), es_kind=int(d["node_term_kind"][i]), expanded=int(d["node_expanded"][i]), ns=int(d["node_ns"][i]))
        lo, n = int(d["node_edge_off"][i]), int(d["node_n_valid"][i])
        rec.update(actions=d["edge_action"][lo:lo + n].astype(np.int64), p=d["edge_p"][lo:lo + n], nsa=d["edge_nsa"][lo:lo + n],
                   q=d["edge_q"][lo:lo + n], q_kind=d["edge_q_kind"][lo:lo + n], child=d["edge_child"][lo:lo + n])
        key = (np.ascontiguousarray(d["node_rows"][i]).tobytes(), np.ascontiguousarray(d["node_rem"][i]).tobytes())
        assert key not in out, "duplicate state in the engine's table"
        out[key] = rec
    return out


def assert_trees_equal(got, want, where=""):
    """got: engine tree_as_dict; want: oracle dump() or fixture dict in the same format.  Bit-exact."""
    assert len(got) == len(want), "%s: %d nodes vs %d" % (where, len(got), len(want))
    for key, w in want.items():
        g = got[key]
        assert g["es"] == w["es"], where
        if w["es"] != 0:
            assert g["es_kind"] == w["es_kind"], where
            continue
        assert g["expanded"] == w["expanded"], where
        if not w["expanded"]:
            continue
        assert g["ns"] == w["ns"], where
        assert np.array_equal(g["actions"], w["actions"]), where
        assert np.array_equal(g["p"], w["p"]), where + " priors differ"
        assert np.array_equal(g["nsa"], w["nsa"]), where
        vis = w["nsa"] > 0
        assert np.array_equal(g["q"][vis], w["q"][vis]), where + " Q differs"
        assert np.array_equal(g["q_kind"][vis], w["q_kind"][vis]), where


def fixture_tree(d):
    """mcts_*.npz -> dict in tree_as_dict format."""
    out = {}
    starts = np.searchsorted(d["e_node"], np.arange(len(d["node_es"]) + 1))
    for i in range(len(d["node_es"])):
        lo, hi = starts[i], starts[i + 1]
        rec = dict(es=int(d["node_es"][i]), es_kind=int(d["node_es_kind"][i]), expanded=int(d["node_exp"][i]), ns=int(d["node_ns"][i]),
                   actions=d["e_act"][lo:hi].astype(np.int64), p=d["e_p"][lo:hi], nsa=d["e_n"][lo:hi], q=d["e_q"][lo:hi], q_kind=d["e_qk"][lo:hi])
        out[(d["node_rows"][i].tobytes(), d["node_rem"][i].tobytes())] = rec
    return out


def planes_from_state(rows, rem, item_wh, W, H):
    """getBinItem planes (N+1, H, W) float32 from a packed state (BinPackingGame.py:45,55,118-120)."""
    N = len(rem)
    out = np.zeros((N + 1, H, W), np.float32)
    out[0] = ev.unpack_board(rows, W)
    for i in range(N):
        if rem[i]:
            out[i + 1, :item_wh[i][1], :item_wh[i][0]] = 1.0
    return out


def tree_digest(tree, N):
    """SHA-256 over a whole search tree in tree_as_dict / OracleMCTS.dump format, covering exactly what assert_trees_equal
    compares: per state (sorted by key) the key, Es and its kind; for live states the expanded flag; for expanded ones Ns and
    every legal move's (action, P, Nsa) plus (Q, kind) of the visited ones.  Returns (hex digest of everything,
    {level: (nodes, hex digest)}) -- level = items placed, so a mismatch can be localised without the other tree at hand.
    A tree too large for the oracle to rebuild inside a GPU test (50x50 / 128 items / 800 sims: ~10^5 nodes, ~5 x 10^7 edges)
    is pinned this way: the digest is generated once from the oracle and committed (tests/golden/make_c5_digest.py)."""
    import hashlib
    import struct
    total = hashlib.sha256()
    levels = {}
    for key in sorted(tree):
        rec = tree[key]
        rem = np.frombuffer(key[1], np.uint8)
        level = int(N - int((rem != 0).sum()))
        parts = [key[0], key[1], struct.pack("<b", int(rec["es"]))]
        if rec["es"] != 0:
            parts.append(struct.pack("<B", int(rec["es_kind"])))
        else:
            parts.append(struct.pack("<B", int(rec["expanded"])))
            if rec["expanded"]:
                nsa = np.ascontiguousarray(rec["nsa"], np.uint32)
                vis = nsa > 0
                parts += [struct.pack("<I", int(rec["ns"])), np.ascontiguousarray(rec["actions"], np.int64).tobytes(),
                          np.ascontiguousarray(rec["p"], np.float64).tobytes(), nsa.tobytes(),
                          np.ascontiguousarray(np.asarray(rec["q"], np.float64)[vis]).tobytes(),
                          np.ascontiguousarray(np.asarray(rec["q_kind"], np.uint8)[vis]).tobytes()]
        lv = levels.setdefault(level, [0, hashlib.sha256()])
        lv[0] += 1
        for p in parts:
            total.update(p); lv[1].update(p)
    return total.hexdigest(), {k: (v[0], v[1].hexdigest()) for k, v in sorted(levels.items())}
