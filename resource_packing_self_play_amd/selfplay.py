"""Batched self-play driver: the many-games-at-once counterpart of `CoachBPP.executeEpisode`
(xw_mcts/CoachBPP.py:50-99) that `CoachBPP.learn` and `bench.py` run.

Every engine slot plays one episode at a time.  The slots of a GPU are split into `groups` (default 2), each with its
own engine context, HIP stream and captured HIP graph of one simulation WAVE:

    rp_search_step      select / descend / terminal backups on device until each slot needs a leaf evaluated
    rp_leaf_stem        leaf states -> first convolution + max-pool of the CNN, computed from the packed state
                        (or rp_leaf_planes: FP32 NCHW planes for a generic evaluator)
    NNetWrapper.predict_from_stem   the rest of the CNN through PyTorch-ROCm (FP32)
    rp_commit_eval      mask / renormalise / expand / backup on device

The groups' waves are launched alternately on their streams, so the latency-bound tree walk of one group runs while the
CNN of the other occupies the matrix cores.  Moves are played on device (`RP_MOVE_SAMPLE` / `RP_MOVE_ARGMAX_FIRST`) and a
finished slot immediately pulls the next instance of its group's pool, so the evaluator batch stays full until the pool
runs dry.
"""
import time

import numpy as np
import torch

from . import _lib


class _Group:
    def __init__(self, owner, index, games, seed):
        self.index = index
        self.G = games
        self.stream = torch.cuda.Stream(owner.device)
        self.eng = _lib.Engine(owner.W, owner.H, owner.N, games, int(owner.args.numMCTSSims), cpuct=float(owner.args.cpuct),
                               alpha=float(owner.args.alpha), move_rule=owner.move_rule, seed=seed,
                               tie_salt=(seed ^ 0x5DEECE66D) if owner.tie_salt is None else int(owner.tie_salt),
                               node_cap=owner.node_cap, edge_cap=owner.edge_cap, device=owner.device.index or 0,
                               stream=self.stream.cuda_stream, auto_restart=1, max_examples=owner.max_examples_per_group,
                               vis_cap=owner.vis_cap, reclaim=1 if owner.reclaim else 0)
        self.eng.set_step_cap(owner.step_cap)
        self.eng.set_compact_rows(owner.compact_rows)
        self.use_stem = owner.use_stem
        self.host_evaluator = owner.host_evaluator
        if self.use_stem:  # the engine computes the first conv + pool itself: no plane tensor at all
            self.channels_last = owner.channels_last
            self.resblock_kernel = owner.resblock_kernel
            fmt = torch.channels_last if self.channels_last else torch.contiguous_format
            self.stem = torch.zeros((games, 16, (owner.H + 1) // 2, (owner.W + 1) // 2), dtype=torch.float32, device=owner.device).contiguous(memory_format=fmt)
            # relu(stem) is only read by the library path of stage 0; k_resstage16 takes x alone and applies the ReLU itself, so with
            # the stage kernels the engine is handed NULL and writes 6.4 KB per leaf instead of 12.8 (half of k_leaf_stem's HBM bytes)
            self.fused = bool(owner.fuse_elementwise)
            stage0_kernel = owner.resblock_kernel and ((owner.H + 1) // 2) * ((owner.W + 1) // 2) <= 640  # BinPackingNNet.STAGE16_MAX_PIXELS
            self.stem_relu = torch.zeros_like(self.stem) if (owner.fuse_elementwise and not stage0_kernel) else None
            self.planes = None
        else:
            self.planes = torch.zeros((games, owner.N + 1, owner.H, owner.W), dtype=torch.float32, device=owner.device)
        self.graph = None
        self.pi = self.v = None
        self.raw_logits = False

    def refresh_weights(self, nnet):
        if self.use_stem:
            w, b = nnet.stem_params()
            self._stem_w, self._stem_b = w, b  # keep alive until the table kernel ran
            self.eng.stem_set_weights(w.data_ptr(), b.data_ptr())
            if self.fused and self.channels_last and self.resblock_kernel:
                self._frag_src = nnet.nnet.refresh_frags(self.eng)

    def leaf_inputs(self):
        """Evaluator input of the waiting leaves: the stem (first convolution + pool from the packed state) or the dense planes."""
        if self.use_stem:
            self.eng.leaf_stem(self.stem.data_ptr(), self.G, self.stem_relu.data_ptr() if self.stem_relu is not None else None, self.channels_last)
        else:
            self.eng.leaf_planes(self.planes.data_ptr(), self.G)

    def forward(self, nnet):
        """-> (policy rows, values).  With the fused evaluator the policy rows are the RAW logits and the softmax is taken inside the
        commit kernel (rp_commit_eval_logits) when the action space fits its LDS row buffer."""
        if self.use_stem:
            self.raw_logits = self.fused and self.eng.A <= self.eng.LOGITS_MAX_ACTIONS
            return nnet.predict_from_stem(self.stem, self.stem_relu, self.eng if self.fused else None, logits=self.raw_logits)
        self.raw_logits = False
        return nnet.predict_batch(self.planes)

    def commit(self, pi, v):
        (self.eng.commit_eval_logits if self.raw_logits else self.eng.commit_eval)(pi.data_ptr(), v.data_ptr())

    def wave_eager(self, nnet):
        if self.host_evaluator is not None:  # generic evaluator on the host: leaf states out, (pi, v) back (like nnet.predict, MCTS_bpp.py:87)
            n = self.eng.search_step()
            if n:
                rows, rem, slots = self.eng.leaf_states(n)
                pi, v = self.host_evaluator(rows, rem, slots)
                self.eng.commit_eval_host(pi, v)
            return
        self.eng.search_step(sync=False)
        self.leaf_inputs()
        self.pi, self.v = self.forward(nnet)
        self.commit(self.pi, self.v)


class BatchedSelfPlay:
    def __init__(self, game, nnet, args, games, move_rule=_lib.MOVE_SAMPLE, seed=0, node_cap=0, edge_cap=0, max_examples=0,
                 use_graph=True, groups=2, step_cap=4, use_stem=True, fuse_elementwise=True, dense_small_convs=True, reclaim=True, vis_cap=0, compact_rows=True, channels_last=True, resblock_kernel=True, device=None,
                 tie_salt=None, host_evaluator=None):
        """host_evaluator: optional callable (rows [n][H] uint64, remaining [n][N] uint8, slots [n]) -> (pi [n][A] float32, v [n]
        float32) that replaces the CNN -- any object with the reference's `predict` contract can sit behind it; the waves then run
        eagerly with one host round trip each.  tie_salt: salt of the deterministic `r == bl` tie (default: derived from seed)."""
        self.game, self.nnet, self.args = game, nnet, args
        self.W, self.H, self.N = game.bin_width, game.bin_height, game.num_items
        self.A = self.W * self.N
        if nnet.device.type != "cuda":
            raise RuntimeError("BatchedSelfPlay needs the evaluator on the GPU (args.cuda = True); there is no CPU path")
        self.device = nnet.device if device is None else device
        groups = max(1, min(int(groups), int(games)))
        self.G = int(games)
        self.move_rule, self.node_cap, self.edge_cap, self.step_cap = move_rule, node_cap, edge_cap, int(step_cap)
        self.reclaim, self.vis_cap = bool(reclaim), int(vis_cap)
        self.compact_rows = bool(compact_rows)  # evaluator rows = the waiting slots only, listed on the device (rp_set_compact_rows)
        self.max_examples_per_group = (int(max_examples) + groups - 1) // groups if max_examples else 0
        self.tie_salt, self.host_evaluator = tie_salt, host_evaluator
        self.use_graph = bool(use_graph) and host_evaluator is None
        self.use_stem = bool(use_stem)
        self.fuse_elementwise = bool(fuse_elementwise) and self.use_stem
        self.dense_small_convs = bool(dense_small_convs)
        self.channels_last = bool(channels_last) and self.fuse_elementwise
        self.resblock_kernel = bool(resblock_kernel) and self.channels_last
        nnet.nnet.use_resblock_kernel = self.resblock_kernel
        if self.channels_last:  # convolution weights in NHWC too, so MIOpen never converts per call
            nnet.nnet.to(memory_format=torch.channels_last)
        sizes = [self.G // groups + (1 if k < self.G % groups else 0) for k in range(groups)]
        # every group (and every rank) samples with the SAME seed: a move's draw depends on (seed, global episode id, move) only,
        # so episodes do not depend on how many groups or ranks share the pool
        self.groups = [_Group(self, k, sizes[k], int(seed) & 0x7FFFFFFFFFFFFFFF) for k in range(groups)]
        self.steps = 0
        self.first_id = 0

    @property
    def eng(self):
        """First group's engine (single-group callers, tests)."""
        return self.groups[0].eng

    def close(self):
        for g in self.groups:
            g.graph = None
            g.pi = g.v = g.stem = g.stem_relu = g.planes = None
            g.eng.close()

    @property
    def device_bytes(self):
        return sum(g.eng.device_bytes for g in self.groups)

    # ---- waves -----------------------------------------------------------------------------------
    def prepare(self):
        """Warms the evaluator up (MIOpen picks its kernels on the first calls) and captures every group's wave -- the
        engine's kernels and the CNN's -- into one HIP graph each, so a wave costs one graph launch instead of ~40 kernel
        launches."""
        if self.host_evaluator is not None:
            return
        if self.fuse_elementwise and self.dense_small_convs:
            self.nnet.refresh_fused()
        if not self.use_graph:
            return
        for g in self.groups:
            if g.graph is not None:
                continue
            torch.cuda.synchronize(self.device)
            with torch.cuda.stream(g.stream):
                g.refresh_weights(self.nnet)
                for _ in range(3):
                    g.forward(self.nnet)
            torch.cuda.synchronize(self.device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=g.stream):
                g.wave_eager(self.nnet)
            torch.cuda.synchronize(self.device)
            g.graph = graph

    def invalidate_graph(self):
        """Call after the evaluator's weights were REPLACED (not updated in place), or after set_move_rule / set_sims:
        kernel arguments are baked into the captured graphs."""
        for g in self.groups:
            g.graph = None

    def step_group(self, g):
        with torch.cuda.stream(g.stream):
            if self.use_graph:
                if g.graph is None:
                    self.prepare()
                g.graph.replay()
            else:
                g.wave_eager(self.nnet)

    def step(self):
        """One simulation wave for every group (each on its own stream)."""
        for g in self.groups:
            self.step_group(g)
        self.steps += 1

    def set_move_rule(self, move_rule, onehot_examples=False):
        for g in self.groups:
            g.eng.set_move_rule(move_rule, onehot_examples)
        self.move_rule = move_rule
        self.invalidate_graph()

    # ---- whole pools -------------------------------------------------------------------------------
    def _blocks(self, n):
        """Contiguous block of the pool per group: [(lo, hi)]; instance i has episode id first_id + i."""
        k = len(self.groups)
        base, rem = divmod(int(n), k)
        out, lo = [], 0
        for g in range(k):
            hi = lo + base + (1 if g < rem else 0)
            out.append((lo, hi)); lo = hi
        return out

    def start(self, item_wh, total_area, rewards_list=(), first_id=0):
        """The pool is cut into one contiguous block per group; instance i's episode id is first_id + i."""
        item_wh = np.ascontiguousarray(item_wh, dtype=np.uint8)
        total_area = np.ascontiguousarray(total_area, dtype=np.int32)
        buf = np.asarray(list(rewards_list), dtype=np.float64)
        self.first_id = int(first_id)
        self.n_instances = item_wh.shape[0]
        torch.cuda.synchronize(self.device)  # the evaluator's weights may just have been trained on another stream
        if self.host_evaluator is None and self.fuse_elementwise and self.dense_small_convs:
            self.nnet.refresh_fused()
            torch.cuda.synchronize(self.device)
        for g, (lo, hi) in zip(self.groups, self._blocks(self.n_instances)):
            wh_g = np.ascontiguousarray(item_wh[lo:hi]); area_g = np.ascontiguousarray(total_area[lo:hi])
            if self.host_evaluator is None:
                with torch.cuda.stream(g.stream):
                    g.refresh_weights(self.nnet)  # stem tables follow in-place weight updates
            g.eng.set_rank_buffer(buf)
            g.eng._ck(g.eng.L.rp_set_instance_pool(g.eng.h, wh_g.shape[0], _lib._ptr(wh_g), _lib._ptr(area_g), self.first_id + lo))
            g.eng._ck(g.eng.L.rp_begin_pool(g.eng.h))

    def start_from_seeds(self, seeds, rewards_list=(), first_id=0, bin_h=None, bin_w=None):
        """Like start(), with instance i = ItemsGenerator.items_generator(seeds[i]) of the bin_w x bin_h rectangle (default: the
        board) generated on the device (rp_set_instance_pool_seeds: bit-identical to the host generator, tests/test_gpu_rules.py);
        total area bin_w * bin_h."""
        seeds = np.ascontiguousarray(seeds, dtype=np.uint32).reshape(-1)
        buf = np.asarray(list(rewards_list), dtype=np.float64)
        self.first_id = int(first_id)
        self.n_instances = seeds.shape[0]
        torch.cuda.synchronize(self.device)
        if self.host_evaluator is None and self.fuse_elementwise and self.dense_small_convs:
            self.nnet.refresh_fused()
            torch.cuda.synchronize(self.device)
        for g, (lo, hi) in zip(self.groups, self._blocks(self.n_instances)):
            if self.host_evaluator is None:
                with torch.cuda.stream(g.stream):
                    g.refresh_weights(self.nnet)
            g.eng.set_rank_buffer(buf)
            g.eng.set_instance_pool_seeds(np.ascontiguousarray(seeds[lo:hi]), bin_w or self.W, bin_h or self.H, self.first_id + lo)
            g.eng._ck(g.eng.L.rp_begin_pool(g.eng.h))

    def active(self):
        n = 0
        for g in self.groups:
            ph, _, _, _ = g.eng.status()
            n += int(np.isin(ph, (_lib.PHASE_RUNNING, _lib.PHASE_WAIT_EVAL, _lib.PHASE_MOVE_READY)).sum())
        return n

    def pop_finished(self):
        """(episode ids, outcomes, scores, moves) of the episodes finished since the last call, sorted by id."""
        parts = []
        for g in self.groups:
            ids, oc, sc, mv = g.eng.pop_finished()
            parts.append((ids.astype(np.int64), oc, sc, mv))
        ids = np.concatenate([p[0] for p in parts]); order = np.argsort(ids, kind="stable")
        return tuple(np.concatenate([p[j] for p in parts])[order] for j in range(4))

    def arena_peak(self):
        peaks = [g.eng.arena_peak() for g in self.groups]
        return {k: max(pk[k] for pk in peaks) for k in peaks[0]}

    def counters(self, reset=False):
        tot = dict.fromkeys(_lib.COUNTER_NAMES, 0)
        for g in self.groups:
            for name, val in g.eng.counters(reset=reset).items():
                tot[name] += val
        return tot

    def run(self, item_wh, total_area, rewards_list=(), first_id=0, poll=16, max_steps=None):
        """Plays every instance of the pool to the end.  Returns (episode ids, outcomes, scores, moves) sorted by id,
        plus timing / counter statistics."""
        self.start(item_wh, total_area, rewards_list, first_id)
        return self._play_out(poll, max_steps)

    def run_from_seeds(self, seeds, rewards_list=(), first_id=0, bin_h=None, poll=16, max_steps=None, bin_w=None):
        """run() on device-generated instances (start_from_seeds)."""
        self.start_from_seeds(seeds, rewards_list, first_id, bin_h, bin_w)
        return self._play_out(poll, max_steps)

    def _play_out(self, poll, max_steps):
        t0 = time.time()
        steps0 = self.steps
        while self.active() > 0:
            for _ in range(poll):
                self.step()
            if max_steps is not None and self.steps - steps0 >= max_steps:
                break
        torch.cuda.synchronize(self.device)
        dt = time.time() - t0
        ids, outcome, score, moves = self.pop_finished()
        stats = self.counters()
        stats.update(seconds=dt, steps=self.steps - steps0, episodes_finished=len(ids))
        return ids, outcome, score, moves, stats

    # ---- replay ---------------------------------------------------------------------------------
    def examples_packed(self):
        """Everything recorded so far as ONE PackedReplay (replay.py: ~0.4 KB per example) in the reference's order: episode by
        episode, move by move (CoachBPP.py:80,133) -- the device buffers themselves fill in completion order across slots."""
        from .replay import PackedReplay
        torch.cuda.synchronize(self.device)
        parts = [PackedReplay.from_engine(g.eng, self.device) for g in self.groups]
        torch.cuda.synchronize(self.device)  # the copies ran on the groups' streams
        return PackedReplay.cat(parts).sort_by_episode_move()

    def examples(self, with_meta=False):
        """(planes [E, N+1, H, W], pi [E, A], value [E]) float32 device tensors of everything recorded so far, in the reference's
        order -- the packed set expanded once (small pools, tests; training expands per minibatch instead).
        with_meta: also (episode ids [E] int64, move numbers [E] int32) as numpy arrays."""
        rep = self.examples_packed()
        res = rep.dense()
        torch.cuda.synchronize(self.device)
        return res + (rep.episode.cpu().numpy(), rep.move.cpu().numpy()) if with_meta else res

    def clear_examples(self):
        for g in self.groups:
            g.eng._ck(g.eng.L.rp_examples_clear(g.eng.h))
