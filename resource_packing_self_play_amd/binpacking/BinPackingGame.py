"""`BinPackingGame` / `ItemsGenerator` with the reference's names and signatures
(xw_mcts/binpacking/BinPackingGame.py:8-285), backed by the HIP engine.

The rule methods (`getValidMoves`, `getNextState`, `has_valid_moves`, `getGameEnded`,
`getRankedReward`) pack the reference's (N+1, H, W) state into row bit masks and call the
C ABI (`rp_valid_moves`, `rp_apply_move`, `rp_game_ended`); there is no Python or CPU
re-implementation of the rules here, so they need the GPU library.  Everything else is
host bookkeeping (array construction, the instance generator).
"""
import numpy as np

from .. import state as st


class BinPackingGame:
    def __init__(self, bin_width, bin_height, num_items, n):
        self.bin_width = int(bin_width)
        self.bin_height = int(bin_height)
        self.num_items = int(num_items)
        self.n = n  # number of bins (always 1 in the reference's runs)
        self.cur_item = 0
        # per-episode state the reference keeps on the game object (BinPackingGame.py:21-22,49-50)
        self.sum_h = 0
        self.max_h = 0
        self._item_wh = None
        self._engine = None

    # ---- plain host bookkeeping ------------------------------------------------------------
    def getInitBoard(self):
        return np.zeros((self.bin_height, self.bin_width), dtype=np.int64)

    def getBoardSize(self):
        return (self.bin_height, self.bin_width)

    def getActionSize(self):
        return self.bin_width * self.num_items

    def getInitItems(self, items_list):
        """items_list rows are [w, h, a, b]; item i becomes a plane of ones in [0:h, 0:w] (reference :37-51)."""
        sizes = np.asarray([list(it)[:2] for it in list(items_list)[:self.num_items]], dtype=np.int64)
        planes = []
        for w, h in sizes:
            plane = self.getInitBoard()
            plane[:h, :w] = 1
            planes.append(plane)
        self.sum_h = int(sizes[:, 1].sum())
        self.max_h = int(sizes[:, 1].max())
        self._item_wh = sizes.copy()
        return planes

    def getItemsUpdated(self, items_list_board, cur_item):
        items_list_board[cur_item] = items_list_board[cur_item] * 0
        return items_list_board

    def getBinItem(self, board, items_list_board):
        return np.array([board] + list(items_list_board))

    def stringRepresentation(self, board):
        return b"".join(np.ascontiguousarray(plane).tobytes() for plane in board)

    # ---- rules on the GPU --------------------------------------------------------------------
    def _eng(self):
        if self._engine is None:
            from .._lib import Engine
            self._engine = Engine(self.bin_width, self.bin_height, self.num_items, games=1, sims=1)
        return self._engine

    def _pack(self, state):
        rows, remaining, wh = st.pack_state(state, self._item_wh if self._item_wh is not None and len(self._item_wh) == len(state) - 1 else None)
        return rows[None], remaining[None], wh[None]

    def getValidMoves(self, board):
        rows, remaining, wh = self._pack(board)
        mask, n_valid = self._eng().valid_moves(rows, remaining, wh)
        assert n_valid[0] > 0  # reference :89
        return mask[0].astype(np.int64)

    def has_valid_moves(self, board):
        rows, remaining, wh = self._pack(board)
        _, n_valid = self._eng().valid_moves(rows, remaining, wh)
        return bool(n_valid[0] > 0)

    def getNextState(self, board, action, items_list_board):
        items = np.copy(items_list_board)
        state = np.concatenate([np.asarray(board)[None], items], axis=0)
        rows, remaining, wh = self._pack(state)
        rows2, _, status = self._eng().apply_move(rows, remaining, wh, [int(action)])
        assert status[0] == 0  # must choose an unplaced item (reference :69)
        item = int(int(action) / self.bin_width)
        items[item] = items[item] * 0
        return (st.unpack_rows(rows2[0], self.bin_width, np.asarray(board).dtype), items)

    def _ranked(self, total_board, items_total_area, rewards_list, alpha, only_if_stuck):
        rows, remaining, wh = self._pack(total_board)
        if not only_if_stuck:
            remaining = np.zeros_like(remaining)  # skip the move test: rank this grid as it stands
        ended, reward = self._eng().game_ended(rows, remaining, wh, [int(items_total_area)], [int(self.max_h)],
                                               np.asarray(list(rewards_list), dtype=np.float64), float(alpha))
        e, r = int(ended[0]), reward[0]
        if e == 2:  # r == bl: the reference draws (reference :211-212)
            e = np.random.choice([1, -1], p=[0.5, 0.5])
        return e, r

    def getGameEnded(self, total_board, items_total_area, rewards_list, alpha):
        assert len(total_board) == self.num_items + self.n
        e, r = self._ranked(total_board, items_total_area, rewards_list, alpha, only_if_stuck=True)
        if e == 0:
            return 0, []
        return e, r

    def getRankedReward(self, total_board, items_total_area, rewards_list, alpha):
        return self._ranked(total_board, items_total_area, rewards_list, alpha, only_if_stuck=False)

    def get_minimal_bin_height(self, board):
        occupied = np.nonzero(np.asarray(board).sum(axis=1) > 0)[0]
        return int(occupied[-1]) + 1 if len(occupied) else 1


class ItemsGenerator:
    """Random guillotine cuts of the bin_width x bin_height rectangle into `items` pieces [w, h, a, b], drawing from
    NumPy's legacy global stream exactly in the reference's order (reference :250-285), so a seed gives the same
    instance.  `bin_height` is mutable: CoachBPP re-draws it every iteration (CoachBPP.py:118)."""

    def __init__(self, bin_width, bin_height, items):
        self.bin_width = bin_width
        self.bin_height = bin_height
        self.n = items

    def items_generator(self, seed):
        np.random.seed(seed)
        pieces = [[self.bin_width, self.bin_height, 0, 0]]
        while len(pieces) < self.n:
            cut_rows = np.random.randint(2) == 1  # axis draw comes first, then the piece index
            k = np.random.randint(len(pieces))
            w, h, a, b = pieces[k]
            if (h if cut_rows else w) == 1:
                continue  # cannot cut a unit side; both draws are already consumed
            if cut_rows:
                cut = np.random.randint(b + 1, b + h) - b
                halves = [[w, cut, a, b], [w, h - cut, a, b + cut]]
            else:
                cut = np.random.randint(a + 1, a + w) - a
                halves = [[cut, h, a, b], [w - cut, h, a + cut, b]]
            del pieces[k]
            pieces.extend(halves)
        return pieces
