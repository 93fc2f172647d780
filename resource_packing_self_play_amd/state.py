"""Conversions between the reference's state tensors and the engine's packed form (host logic).

Reference state: int64 array (N+1, H, W) -- plane 0 the grid, plane i+1 item i drawn as ones in [0:h, 0:w]
while unplaced, all zeros once placed (BinPackingGame.py:45,55,118-120).
Engine state:    rows uint64[H] (bit c = cell (r, c)), remaining uint8[N], item sizes uint8[N][2] = (w, h).
"""
import numpy as np


def pack_rows(board):
    board = np.asarray(board)
    if board.ndim != 2:
        raise ValueError("board must be (H, W)")
    if board.size and not np.isin(board, (0, 1)).all():
        raise ValueError("the packed engine needs 0/1 grids")
    weights = np.uint64(1) << np.arange(board.shape[1], dtype=np.uint64)
    return (board.astype(np.uint64) * weights[None, :]).sum(axis=1, dtype=np.uint64)


def unpack_rows(rows, width, dtype=np.int64):
    rows = np.asarray(rows, dtype=np.uint64)
    return ((rows[:, None] >> np.arange(width, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(dtype)


def item_sizes(items_planes):
    """(w, h) of every item plane exactly as the reference reads them: w = sum(item[0, :]), h = sum(item[:, 0])
    (BinPackingLogic.py:84-85); remaining = plane sum != 0 (BinPackingGame.py:86)."""
    items = np.asarray(items_planes)
    remaining = (items.reshape(items.shape[0], -1).sum(axis=1) != 0).astype(np.uint8)
    wh = np.stack([items[:, 0, :].sum(axis=1), items[:, :, 0].sum(axis=1)], axis=1)
    return wh, remaining


def pack_state(state, known_wh=None):
    """state (N+1, H, W) -> rows, remaining, item_wh (uint8).  A placed item's plane is all zero, so its size cannot be
    read back from the state; `known_wh` (from getInitItems) fills those in, otherwise they are set to 1x1 (never read)."""
    state = np.asarray(state)
    rows = pack_rows(state[0])
    wh, remaining = item_sizes(state[1:])
    wh = wh.astype(np.int64)
    if known_wh is not None:
        wh = np.where(remaining[:, None] != 0, wh, np.asarray(known_wh, dtype=np.int64))
    wh = np.where(wh == 0, 1, wh)
    return rows, remaining, wh.astype(np.uint8)


def unpack_state(rows, remaining, item_wh, width, height):
    """inverse of pack_state -> int64 (N+1, H, W)"""
    n = len(remaining)
    out = np.zeros((n + 1, height, width), dtype=np.int64)
    out[0] = unpack_rows(rows, width)
    for i in range(n):
        if remaining[i]:
            out[i + 1, :int(item_wh[i][1]), :int(item_wh[i][0])] = 1
    return out
