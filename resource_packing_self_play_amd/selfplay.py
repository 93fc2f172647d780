"""Batched self-play driver: the many-games-at-once counterpart of `CoachBPP.executeEpisode`
(xw_mcts/CoachBPP.py:50-99) that `CoachBPP.learn` and `bench.py` run.

Every engine slot plays one episode at a time; all slots advance in lock step, one evaluator call per step:

    rp_search_step      select / descend / terminal backups on device until each slot needs a leaf evaluated
    rp_leaf_planes      leaf states -> FP32 NCHW planes written straight into the evaluator's input tensor
    NNetWrapper.predict_batch   the CNN through PyTorch-ROCm (FP32, optionally replayed from a captured HIP graph)
    rp_commit_eval      mask / renormalise / expand / backup on device

Moves are played on device (`RP_MOVE_SAMPLE` or `RP_MOVE_ARGMAX_FIRST`) and a finished slot immediately pulls the
next instance of the pool, so the evaluator batch stays full until the pool runs dry.
"""
import time

import numpy as np
import torch

from . import _lib


class BatchedSelfPlay:
    def __init__(self, game, nnet, args, games, move_rule=_lib.MOVE_SAMPLE, seed=0, node_cap=0, edge_cap=0, max_examples=0,
                 use_graph=True, device=None):
        self.game, self.nnet, self.args = game, nnet, args
        self.W, self.H, self.N = game.bin_width, game.bin_height, game.num_items
        self.A = self.W * self.N
        self.G = int(games)
        if nnet.device.type != "cuda":
            raise RuntimeError("BatchedSelfPlay needs the evaluator on the GPU (args.cuda = True); there is no CPU path")
        self.device = nnet.device if device is None else device
        self.stream = torch.cuda.current_stream(self.device)
        self.eng = _lib.Engine(self.W, self.H, self.N, self.G, int(args.numMCTSSims), cpuct=float(args.cpuct), alpha=float(args.alpha),
                               move_rule=move_rule, seed=seed, tie_salt=seed ^ 0x5DEECE66D, node_cap=node_cap, edge_cap=edge_cap,
                               device=self.device.index or 0, stream=self.stream.cuda_stream, auto_restart=1, max_examples=max_examples)
        self.planes = torch.zeros((self.G, self.N + 1, self.H, self.W), dtype=torch.float32, device=self.device)
        self.use_graph = use_graph
        self._graph = None
        self._pi = self._v = None
        self.steps = 0

    def close(self):
        self.eng.close()

    # ---- one simulation wave: search -> planes -> CNN -> commit, static shapes, replayed from one HIP graph -----------
    def _wave_eager(self):
        self.eng.search_step(sync=False)
        self.eng.leaf_planes(self.planes.data_ptr(), self.G)
        self._pi, self._v = self.nnet.predict_batch(self.planes)
        self.eng.commit_eval(self._pi.data_ptr(), self._v.data_ptr())

    def prepare(self):
        """Warms the evaluator up (MIOpen picks its kernels on the first calls) and captures the whole wave -- the engine's
        kernels and the CNN's -- into one HIP graph, so a wave costs one graph launch instead of ~40 kernel launches."""
        if self._graph is not None or not self.use_graph:
            return
        side = torch.cuda.Stream(self.device)
        side.wait_stream(self.stream)
        with torch.cuda.stream(side):
            for _ in range(3):
                self.nnet.predict_batch(self.planes)
        self.stream.wait_stream(side)
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        self.eng.set_stream(side.cuda_stream)
        try:
            with torch.cuda.graph(graph, stream=side):
                self._wave_eager()
        finally:
            self.eng.set_stream(self.stream.cuda_stream)
        torch.cuda.synchronize(self.device)
        self._graph = graph

    def invalidate_graph(self):
        """Call after the evaluator's weights were REPLACED (not updated in place), or after set_move_rule / set_sims:
        kernel arguments are baked into the captured graph."""
        self._graph = None

    def step(self):
        """One lock-step simulation wave for every slot."""
        if self.use_graph:
            if self._graph is None:
                self.prepare()
            self._graph.replay()
        else:
            self._wave_eager()
        self.steps += 1

    # ---- whole pools ---------------------------------------------------------------------------
    def start(self, item_wh, total_area, rewards_list=(), first_id=0):
        self.eng.set_rank_buffer(np.asarray(list(rewards_list), dtype=np.float64))
        item_wh = np.ascontiguousarray(item_wh, dtype=np.uint8)
        total_area = np.ascontiguousarray(total_area, dtype=np.int32)
        self.eng._ck(self.eng.L.rp_set_instance_pool(self.eng.h, item_wh.shape[0], _lib._ptr(item_wh), _lib._ptr(total_area), int(first_id)))
        self.eng._ck(self.eng.L.rp_begin_pool(self.eng.h))
        self.n_instances = item_wh.shape[0]

    def active(self):
        ph, _, _, _ = self.eng.status()
        return int(np.isin(ph, (_lib.PHASE_RUNNING, _lib.PHASE_WAIT_EVAL, _lib.PHASE_MOVE_READY)).sum())

    def run(self, item_wh, total_area, rewards_list=(), first_id=0, poll=16, max_steps=None):
        """Plays every instance of the pool to the end.  Returns (episode ids, outcomes, scores, moves) sorted by id,
        plus timing / counter statistics."""
        self.start(item_wh, total_area, rewards_list, first_id)
        t0 = time.time()
        steps0 = self.steps
        while self.active() > 0:
            for _ in range(poll):
                self.step()
            if max_steps is not None and self.steps - steps0 >= max_steps:
                break
        torch.cuda.synchronize(self.device)
        dt = time.time() - t0
        ids, outcome, score, moves = self.eng.pop_finished()
        order = np.argsort(ids, kind="stable")
        stats = self.eng.counters()
        stats.update(seconds=dt, steps=self.steps - steps0, episodes_finished=len(ids))
        return ids[order], outcome[order], score[order], moves[order], stats

    # ---- replay ---------------------------------------------------------------------------------
    def examples(self):
        """(planes [E, N+1, H, W], pi [E, A], value [E]) float32 device tensors of everything recorded so far."""
        n = _lib._i64(0)
        self.eng._ck(self.eng.L.rp_examples_count(self.eng.h, _lib.C.byref(n)))
        e = n.value
        planes = torch.empty((e, self.N + 1, self.H, self.W), dtype=torch.float32, device=self.device)
        pi = torch.empty((e, self.A), dtype=torch.float32, device=self.device)
        value = torch.empty((e,), dtype=torch.float32, device=self.device)
        if e:
            self.eng._ck(self.eng.L.rp_examples_tensors(self.eng.h, 0, e, _lib.C.c_void_p(planes.data_ptr()), _lib.C.c_void_p(pi.data_ptr()),
                                                        _lib.C.c_void_p(value.data_ptr())))
        return planes, pi, value

    def clear_examples(self):
        self.eng._ck(self.eng.L.rp_examples_clear(self.eng.h))
