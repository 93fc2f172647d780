"""Deterministic table-driven evaluators and state packing shared by the golden
generator (tests/golden/make_golden.py), the oracle tests and the GPU parity
tests.  Test infrastructure: integer arithmetic only, so every side (Python
reference, C oracle via callback, HIP engine via host-side leaf evaluation)
computes bit-identical (pi, v) for a state.

A state is identified by its packed key: rows[H] (uint64, bit c = cell (r, c))
and the remaining-item flags (uint8[N]).
"""
import numpy as np

M64 = (1 << 64) - 1


def splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M64
    return x ^ (x >> 31)


def pack_board(board):
    """(H, W) 0/1 array -> uint64[H] row bitmasks (bit c = column c)."""
    board = np.asarray(board)
    h, w = board.shape
    weights = (np.uint64(1) << np.arange(w, dtype=np.uint64))
    return (board.astype(np.uint64) * weights[None, :]).sum(axis=1, dtype=np.uint64)


def unpack_board(rows, w):
    rows = np.asarray(rows, dtype=np.uint64)
    return ((rows[:, None] >> np.arange(w, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(np.uint8)


def pack_state(state):
    """Reference state (N+1, H, W) int64 planes -> (rows u64[H], remaining u8[N], item_w u8[N], item_h u8[N]).
    w = sum(item[0, :]), h = sum(item[:, 0]) exactly as BinPackingLogic.py:84-85 derives them;
    for a placed (all-zero) plane both are 0."""
    state = np.asarray(state)
    rows = pack_board(state[0])
    items = state[1:]
    remaining = (items.reshape(items.shape[0], -1).sum(axis=1) != 0).astype(np.uint8)
    iw = items[:, 0, :].sum(axis=1).astype(np.uint8)
    ih = items[:, :, 0].sum(axis=1).astype(np.uint8)
    return rows, remaining, iw, ih


def remaining_words(remaining):
    """uint8[N] flags -> list of python ints, 64 items per word (bit i%64 of word i//64)."""
    remaining = np.asarray(remaining).astype(bool)
    words = []
    for base in range(0, len(remaining), 64):
        w = 0
        for i, f in enumerate(remaining[base:base + 64]):
            if f:
                w |= 1 << i
        words.append(w)
    return words


def key_hash(rows, remaining, salt=0):
    h = splitmix64(salt & M64)
    for r in np.asarray(rows, dtype=np.uint64).tolist():
        h = splitmix64(h ^ int(r))
    for w in remaining_words(remaining):
        h = splitmix64(h ^ w)
    return h


def tie_value(rows, remaining, salt=0):
    """Deterministic stand-in for np.random.choice([1, -1]) (BinPackingGame.py:212)."""
    return 1 if (splitmix64(key_hash(rows, remaining, salt) ^ 0x7469) & 1) else -1


def _stream(h, n):
    """n pseudo-random uint64 derived from h (vectorised splitmix64 over h + i*golden)."""
    with np.errstate(over="ignore"):
        x = np.uint64(h) + np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def table_eval(kind, rows, remaining, action_size, salt=0):
    """-> (pi float32[A], v float32[1]) like NNetWrapper.predict (NNet.py:85).

    kind 'uniform': pi = 1/A, v = 0.
    kind 'hashed' : pi[a] = k/2^24 (k in 1..2^24), v = (k' - 2^23)/2^23: arbitrary f32 values, so the
                    renormalisation exercises NumPy's pairwise-sum order and f64 rounding.
    kind 'sparse' : like 'hashed' with about half the entries exactly 0, and all-zero pi for one state
                    in eight (drives the uniform-over-valids fallback of MCTS_bpp.py:93-100).
    kind 'peaked' : one dominant action, rest tiny (deep narrow trees like a trained net).
    """
    A = int(action_size)
    if kind == "uniform":
        return np.full(A, np.float32(1.0) / np.float32(A), dtype=np.float32), np.zeros(1, dtype=np.float32)
    h = key_hash(rows, remaining, salt)
    x = _stream(h, A + 1)
    k = (x[:A] >> np.uint64(40)).astype(np.int64) + 1
    pi = k.astype(np.float32) * np.float32(2.0 ** -24)
    kv = int(x[A] >> np.uint64(40)) - (1 << 23)
    v = np.array([np.float32(kv) * np.float32(2.0 ** -23)], dtype=np.float32)
    if kind == "hashed":
        return pi, v
    if kind == "sparse":
        keep = ((x[:A] >> np.uint64(8)) & np.uint64(1)).astype(bool)
        pi = np.where(keep, pi, np.float32(0)).astype(np.float32)
        if (h & 7) == 0:
            pi = np.zeros(A, dtype=np.float32)
        return pi, v
    if kind == "peaked":
        pi = (pi * np.float32(2.0 ** -10)).astype(np.float32)
        # boost a block of actions so that whichever of them is valid dominates
        start = int(h % A)
        idx = (start + np.arange(max(1, A // 8))) % A
        pi[idx] = pi[idx] * np.float32(2.0 ** 10)
        return pi, v
    raise ValueError(kind)


EVAL_KINDS = ("uniform", "hashed", "sparse", "peaked")
