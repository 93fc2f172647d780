"""ctypes binding of the parity oracle (oracle/rp_oracle.c).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "_build", "librp_oracle.so")

EVAL_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_float))
TIE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8))

_u8p = C.POINTER(C.c_uint8)
_lib = None


def build(force=False):
    src_newer = (not os.path.exists(LIB_PATH)) or any(
        os.path.getmtime(os.path.join(ORACLE_DIR, f)) > os.path.getmtime(LIB_PATH) for f in ("rp_oracle.c", "rp_oracle.h"))
    if force or src_newer:
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.orc_np_sum_f64.restype = C.c_double
        L.orc_np_sum_f64.argtypes = [C.POINTER(C.c_double), C.c_int64]
        L.orc_valid_moves.argtypes = [C.c_int, C.c_int, C.c_int, _u8p, _u8p, _u8p, _u8p, _u8p]
        L.orc_has_valid_moves.argtypes = [C.c_int, C.c_int, C.c_int, _u8p, _u8p, _u8p, _u8p]
        L.orc_next_state.argtypes = [C.c_int, C.c_int, C.c_int, _u8p, _u8p, _u8p, _u8p, C.c_int]
        L.orc_ranked_reward.argtypes = [C.c_int, C.c_int, _u8p, C.c_int64, C.c_int, C.POINTER(C.c_double), C.c_int, C.c_double,
                                        C.POINTER(C.c_double)]
        L.orc_game_ended.argtypes = [C.c_int, C.c_int, C.c_int, _u8p, _u8p, _u8p, _u8p, C.c_int64, C.c_int,
                                     C.POINTER(C.c_double), C.c_int, C.c_double, C.POINTER(C.c_double)]
        L.orc_q_update.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_uint32, C.c_double, C.c_int]
        L.orc_q_update.restype = None
        L.orc_mcts_new.restype = C.c_void_p
        L.orc_mcts_new.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, EVAL_FN, C.c_void_p, TIE_FN, C.c_void_p]
        L.orc_mcts_free.argtypes = [C.c_void_p]
        L.orc_mcts_free.restype = None
        L.orc_mcts_begin_episode.argtypes = [C.c_void_p, _u8p, _u8p, C.c_int64, C.POINTER(C.c_double), C.c_int]
        L.orc_mcts_begin_episode.restype = None
        L.orc_mcts_action_counts.argtypes = [C.c_void_p, _u8p, _u8p, C.c_int, C.POINTER(C.c_uint32)]
        L.orc_mcts_num_nodes.restype = C.c_int64
        L.orc_mcts_num_nodes.argtypes = [C.c_void_p]
        L.orc_mcts_get_node.argtypes = [C.c_void_p, C.c_int64, _u8p, _u8p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                        C.POINTER(C.c_uint32), _u8p, C.POINTER(C.c_double), C.POINTER(C.c_uint32),
                                        C.POINTER(C.c_double), _u8p]
        L.orc_mcts_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        L.orc_mcts_stats.restype = None
        L.orc_play_episode.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.POINTER(C.c_int32),
                                       C.POINTER(C.c_uint32), C.POINTER(C.c_int), C.POINTER(C.c_double)]
        L.orc_sample_u64.restype = C.c_uint64
        L.orc_sample_u64.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
        _lib = L
    return _lib


def _p8(a):
    return a.ctypes.data_as(_u8p)


def _c8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def np_sum(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return lib().orc_np_sum_f64(a.ctypes.data_as(C.POINTER(C.c_double)), a.size)


def valid_moves(W, H, N, board, iw, ih, rem):
    board, iw, ih, rem = _c8(board), _c8(iw), _c8(ih), _c8(rem)
    out = np.zeros(W * N, np.uint8)
    n = lib().orc_valid_moves(W, H, N, _p8(board), _p8(iw), _p8(ih), _p8(rem), _p8(out))
    return out, n


def has_valid_moves(W, H, N, board, iw, ih, rem):
    board, iw, ih, rem = _c8(board), _c8(iw), _c8(ih), _c8(rem)
    return bool(lib().orc_has_valid_moves(W, H, N, _p8(board), _p8(iw), _p8(ih), _p8(rem)))


def next_state(W, H, N, board, iw, ih, rem, action):
    board, rem = _c8(board).copy(), _c8(rem).copy()
    iw, ih = _c8(iw), _c8(ih)
    rc = lib().orc_next_state(W, H, N, _p8(board), _p8(iw), _p8(ih), _p8(rem), int(action))
    return rc, board, rem


def ranked_reward(W, H, board, area, max_h, buf, alpha):
    board = _c8(board)
    buf = np.ascontiguousarray(buf, dtype=np.float64)
    r = C.c_double()
    ranked = lib().orc_ranked_reward(W, H, _p8(board), int(area), int(max_h), buf.ctypes.data_as(C.POINTER(C.c_double)), len(buf),
                                     float(alpha), C.byref(r))
    return ranked, r.value


def q_update(q, q_kind, n, v, v_kind):
    cq = C.c_double(q); ck = C.c_int(q_kind)
    lib().orc_q_update(C.byref(cq), C.byref(ck), n, float(v), int(v_kind))
    return cq.value, ck.value


# kinds: golden files use 0 weak / 1 f32 / 2 strong, the same as ORC_WEAK / ORC_F32 / ORC_F64


class OracleMCTS:
    """MCTS of the oracle with Python callbacks for evaluator and tie rule (state arrives as cell grids)."""

    def __init__(self, W, H, N, cpuct, alpha, eval_py, tie_py=None):
        self.W, self.H, self.N, self.A = W, H, N, W * N
        self.eval_py, self.tie_py = eval_py, tie_py
        self.evals = 0

        def _eval(user, board, rem, pi, v):
            b = np.ctypeslib.as_array(board, shape=(H, W)); r = np.ctypeslib.as_array(rem, shape=(N,))
            p, val = self.eval_py(b, r)
            np.ctypeslib.as_array(pi, shape=(self.A,))[:] = p
            v[0] = float(np.asarray(val).reshape(-1)[0])
            self.evals += 1

        def _tie(user, board, rem):
            b = np.ctypeslib.as_array(board, shape=(H, W)); r = np.ctypeslib.as_array(rem, shape=(N,))
            return int(self.tie_py(b, r)) if self.tie_py else 1

        self._eval_cb, self._tie_cb = EVAL_FN(_eval), TIE_FN(_tie)
        self.h = lib().orc_mcts_new(W, H, N, float(cpuct), float(alpha), self._eval_cb, None, self._tie_cb, None)

    def close(self):
        if self.h:
            lib().orc_mcts_free(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def begin_episode(self, iw, ih, total_area, buf):
        iw, ih = _c8(iw), _c8(ih)
        buf = np.ascontiguousarray(buf, dtype=np.float64)
        lib().orc_mcts_begin_episode(self.h, _p8(iw), _p8(ih), int(total_area), buf.ctypes.data_as(C.POINTER(C.c_double)), len(buf))

    def action_counts(self, board, rem, sims):
        board, rem = _c8(board), _c8(rem)
        counts = np.zeros(self.A, np.uint32)
        lib().orc_mcts_action_counts(self.h, _p8(board), _p8(rem), int(sims), counts.ctypes.data_as(C.POINTER(C.c_uint32)))
        return counts

    def play_episode(self, sims, policy=0, seed=0, episode_id=0, want_counts=True):
        actions = np.zeros(self.N + 1, np.int32)
        counts = np.zeros((self.N + 1, self.A), np.uint32) if want_counts else None
        outcome = C.c_int(0); score = C.c_double(0)
        moves = lib().orc_play_episode(self.h, int(sims), int(policy), int(seed), int(episode_id),
                                       actions.ctypes.data_as(C.POINTER(C.c_int32)),
                                       counts.ctypes.data_as(C.POINTER(C.c_uint32)) if want_counts else None,
                                       C.byref(outcome), C.byref(score))
        return actions[:moves], (counts[:moves] if want_counts else None), outcome.value, score.value

    def stats(self):
        out = np.zeros(8, np.int64)
        lib().orc_mcts_stats(self.h, out.ctypes.data_as(C.POINTER(C.c_int64)))
        return dict(zip(("searches", "expansions", "terminal_returns", "path_edges", "sum_valid_select", "sum_valid_leaf",
                         "transposition_hits", "nodes"), out.tolist()))

    def dump(self):
        """-> dict keyed by (rows bytes, rem bytes) of node records."""
        from evaluators import pack_board
        n = lib().orc_mcts_num_nodes(self.h)
        W, H, N, A = self.W, self.H, self.N, self.A
        board = np.zeros(H * W, np.uint8); rem = np.zeros(N, np.uint8)
        valids = np.zeros(A, np.uint8); p = np.zeros(A, np.float64); nsa = np.zeros(A, np.uint32)
        q = np.zeros(A, np.float64); qk = np.zeros(A, np.uint8)
        es = C.c_int(); esk = C.c_int(); exp = C.c_int(); ns = C.c_uint32()
        out = {}
        for i in range(n):
            lib().orc_mcts_get_node(self.h, i, _p8(board), _p8(rem), C.byref(es), C.byref(esk), C.byref(exp), C.byref(ns), _p8(valids),
                                    p.ctypes.data_as(C.POINTER(C.c_double)), nsa.ctypes.data_as(C.POINTER(C.c_uint32)),
                                    q.ctypes.data_as(C.POINTER(C.c_double)), _p8(qk))
            rows = pack_board(board.reshape(H, W))
            rec = dict(es=es.value, es_kind=esk.value, expanded=exp.value, ns=ns.value)
            if exp.value:
                idx = np.nonzero(valids)[0]
                rec.update(actions=idx.copy(), p=p[idx].copy(), nsa=nsa[idx].copy(), q=q[idx].copy(), q_kind=qk[idx].copy())
            out[(rows.tobytes(), rem.tobytes())] = rec
        return out
