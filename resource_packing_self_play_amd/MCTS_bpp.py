"""`MCTS` with the reference's interface (xw_mcts/MCTS_bpp.py:11-139) on top of the HIP engine.

One `MCTS` object owns one engine slot: the tree lives in HBM and persists across `getActionProb` calls exactly as
the reference's dicts persist across the moves of an episode (CoachBPP.py:124 builds a new MCTS per episode).
select / expand / backup run in `rp_search_step` and `rp_commit_eval`; the only host work per simulation is the
evaluator call (`nnet.predict`, MCTS_bpp.py:87).  There is no Python search fallback.
"""
import logging

import numpy as np

from . import _lib
from . import state as st

EPS = 1e-8
log = logging.getLogger(__name__)


class MCTS:
    def __init__(self, game, nnet, args):
        self.game = game
        self.nnet = nnet
        self.args = args
        self._eng = None
        self._wh = None
        self._area = None
        self._buf = None
        self._generation = 0    # bumped by every search: the dict views below are rebuilt at most once per generation
        self._dict_cache = None

    # ---- engine plumbing ---------------------------------------------------------------------
    def _engine_for(self, rows, remaining, wh, total_area, rewards):
        g = self.game
        wh_full = wh
        if self._eng is None:
            self._eng = _lib.Engine(g.bin_width, g.bin_height, g.num_items, games=1, sims=int(self.args.numMCTSSims),
                                    cpuct=float(self.args.cpuct), alpha=float(self.args.alpha), move_rule=_lib.MOVE_EXTERNAL,
                                    seed=int(getattr(self.args, "seed", 0) or 0),
                                    tie_salt=int(np.random.randint(1 << 31)),  # the reference's tie draw is random too
                                    node_cap=int(getattr(self.args, "node_cap", 0) or 0) or int(self.args.numMCTSSims) * (g.num_items + 1) + 2)
        if self._wh is None:
            known = getattr(g, "_item_wh", None)
            if known is not None and len(known) == g.num_items:
                wh_full = np.where(remaining[:, None] != 0, wh, np.asarray(known)).astype(np.uint8)
            self._eng.set_rank_buffer(rewards)
            self._buf = list(rewards)
            self._eng.begin_episodes(wh_full[None], [int(total_area)])
            self._wh, self._area = wh_full, int(total_area)
        elif list(rewards) != self._buf:
            self._eng.set_rank_buffer(rewards)
            self._buf = list(rewards)
        self._eng.set_roots(rows[None], remaining[None])
        return self._eng

    def _evaluate_pending(self, eng):
        rows, rem, _ = eng.leaf_states(1)
        state = st.unpack_state(rows[0], rem[0], self._wh, self.game.bin_width, self.game.bin_height)
        pi, v = self.nnet.predict(state)
        eng.commit_eval_host(np.asarray(pi, dtype=np.float32)[None], np.asarray(v, dtype=np.float32).reshape(1))

    def _run(self, canonicalBoard, totalArea, rewardsList, sims):
        rows, remaining, wh = st.pack_state(canonicalBoard, self._wh)
        eng = self._engine_for(rows, remaining, wh, totalArea, rewardsList)
        self._generation += 1
        eng.set_sims(sims)
        while eng.search_step() > 0:
            self._evaluate_pending(eng)
        return eng

    # ---- reference API -----------------------------------------------------------------------
    def getActionProb(self, canonicalBoard, totalArea, rewardsList, greedy_a=1):
        eng = self._run(canonicalBoard, totalArea, rewardsList, int(self.args.numMCTSSims))
        counts = [int(c) for c in eng.root_counts()[0]]
        if greedy_a == 0:
            bestAs = np.array(np.argwhere(counts == np.max(counts))).flatten()
            bestA = np.random.choice(bestAs)
            probs = [0] * len(counts)
            probs[bestA] = 1
            return probs
        counts = [x ** (1. / greedy_a) for x in counts]
        counts_sum = float(sum(counts))
        return [x / counts_sum for x in counts]

    def search(self, canonicalBoard, totalArea, rewardsList):
        """One simulation from `canonicalBoard`; returns the value that was backed up (MCTS_bpp.py:83,104,139)."""
        eng = self._run(canonicalBoard, totalArea, rewardsList, 1)
        v, kind = eng.last_values()
        if kind[0] == _lib.KIND_F32:
            return np.array([v[0]], dtype=np.float32)
        return int(v[0])

    # ---- the reference's public dicts, rebuilt from the device tree on demand -----------------------------------------
    def _tree(self):
        if self._eng is None:
            return None
        return self._eng.dump_tree(0)

    def _state_key(self, rows, rem):
        s = st.unpack_state(rows, rem, self._wh, self.game.bin_width, self.game.bin_height)
        return self.game.stringRepresentation(s)

    def _dicts(self):
        """All six dicts from ONE dump of the device tree, cached until the next search (`len(m.Ns)` then `len(m.Es)` is one dump)."""
        if self._dict_cache is not None and self._dict_cache[0] == self._generation:
            return self._dict_cache[1]
        out = self._build_dicts()
        self._dict_cache = (self._generation, out)
        return out

    def _build_dicts(self):
        out = {k: {} for k in ("Qsa", "Nsa", "Ns", "Ps", "Es", "Vs")}
        d = self._tree()
        if d is None:
            return out
        A = self.game.getActionSize()
        for i in range(len(d["node_term"])):
            s = self._state_key(d["node_rows"][i], d["node_rem"][i])
            out["Es"][s] = int(d["node_term"][i])
            if not d["node_expanded"][i]:
                continue
            lo, n = int(d["node_edge_off"][i]), int(d["node_n_valid"][i])
            acts = d["edge_action"][lo:lo + n].astype(np.int64)
            P = np.zeros(A, np.float64); P[acts] = d["edge_p"][lo:lo + n]
            V = np.zeros(A, np.int64); V[acts] = 1
            out["Ps"][s], out["Vs"][s], out["Ns"][s] = P, V, int(d["node_ns"][i])
            for k, a in enumerate(acts):
                if d["edge_nsa"][lo + k] > 0:
                    q = d["edge_q"][lo + k]
                    out["Qsa"][(s, int(a))] = np.array([q], dtype=np.float32) if d["edge_q_kind"][lo + k] == _lib.KIND_F32 else float(q)
                    out["Nsa"][(s, int(a))] = int(d["edge_nsa"][lo + k])
        return out

    Qsa = property(lambda self: self._dicts()["Qsa"])
    Nsa = property(lambda self: self._dicts()["Nsa"])
    Ns = property(lambda self: self._dicts()["Ns"])
    Ps = property(lambda self: self._dicts()["Ps"])
    Es = property(lambda self: self._dicts()["Es"])
    Vs = property(lambda self: self._dicts()["Vs"])

    def close(self):
        if self._eng is not None:
            self._eng.close()
            self._eng = None
        self._dict_cache = None
