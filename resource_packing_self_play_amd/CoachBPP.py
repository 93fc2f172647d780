"""`CoachBPP` with the reference's constructor and methods (xw_mcts/CoachBPP.py:22-231): `learn()`,
`executeEpisode(greedy=False)`, `save_rewards_list`, `saveTrainExamples`, `loadTrainExamples`, attributes
`rewards_list`, `ep_score`, `trainExamplesHistory`.

`learn()` is the batched counterpart of the reference loop: the `numEps` episodes of an iteration are independent games
(CoachBPP.py:123-134), so they run concurrently through `BatchedSelfPlay`, sharded over the ranks of a
`torch.distributed` job.  Documented differences from the reference, all forced by concurrency or by its use of OS
entropy:
  * the ranked-reward buffer is a snapshot per iteration: every episode of an iteration is ranked against the buffer as
    it stood when the iteration began; scores are appended in episode order afterwards (the reference appends after
    each sequential episode, CoachBPP.py:134);
  * moves are sampled with the engine's counter-based RNG instead of `np.random.seed(); np.random.choice`
    (CoachBPP.py:86-87): the draw of (episode, move) is a function of (sample seed, running episode number, move), the running
    number counts every episode this Coach has played (so iterations never reuse a stream) and the sample seed comes from OS
    entropy like the reference's draws unless `args.sample_seed` pins it; greedy ties go to the lowest action instead of a
    random one (MCTS_bpp.py:45-46);
  * the replay set is kept PACKED (replay.PackedReplay: state key, item sizes, sparse visit counts -- ~0.4 KB per example
    instead of the reference's 110 KB of int64 planes + float list) and expanded to planes / pi per training minibatch;
  * the per-episode generator seeds are drawn up front from one OS-seeded stream (`drawIteration`);
  * with several ranks a training step still covers `batch_size` examples: one index stream shared by all ranks, each rank
    takes every world-th index, gradients are summed (see `NNetWrapper.train_tensors`).
`executeEpisode` keeps the reference's sequential semantics through the `MCTS` / `BinPackingGame` classes.
"""
import logging
import os
import pickle

import numpy as np

from . import _lib
from . import distributed as rdist
from .MCTS_bpp import MCTS

log = logging.getLogger(__name__)


def _opt(args, name, default):
    """Optional attribute of an args object; the reference's own dotdict (utils.py:20-22) raises KeyError, not AttributeError."""
    try:
        return getattr(args, name)
    except (AttributeError, KeyError):
        return default


def trim_min(scores, cap):
    """`while len(l) > cap: l.pop(argmin(l))` (CoachBPP.py:136-139) in one pass: the loop removes the len - cap smallest values, the
    EARLIEST one first among equals (argmin returns the first minimum), and keeps the order of the rest.  With 32 768 episodes per
    iteration the literal loop is 32 768 scans of a 32 868-element list."""
    scores = list(scores)
    k = len(scores) - int(cap)
    if k <= 0:
        return scores
    a = np.asarray(scores, dtype=np.float64)
    drop = np.lexsort((np.arange(len(a)), a))[:k]  # ascending value, ties by position
    keep = np.ones(len(a), dtype=bool)
    keep[drop] = False
    return [s for s, m in zip(scores, keep) if m]


class CoachBPP:
    def __init__(self, game, nnet, items_list, total_area, gen, args, saved_rewards_list=[]):
        self.game = game
        self.nnet = nnet
        self.args = args
        self.items_list = items_list
        self.items_total_area = total_area
        self.rewards_list = list(saved_rewards_list)
        self.ep_score = 0
        self.mcts = MCTS(self.game, self.nnet, self.args)
        self.trainExamplesHistory = []  # one PackedReplay per iteration (device tensors; .dense() gives planes / pi / value)
        self.episodes_played = 0  # running episode number: the base of an iteration's global episode ids (sampling streams)
        self.timings = []  # per iteration: self-play seconds, replay exchange bytes / ms, training seconds and steps
        self.skipFirstSelfPlay = False
        self.gen = gen
        self.metrics_log = []  # dicts with the reference's W&B metric names, one per iteration
        self.iteration_scores = []  # ep_scores of every iteration, in episode order
        self._selfplay = None

    # ---- sequential episode, reference semantics (CoachBPP.py:50-99) --------------------------------------------------
    def executeEpisode(self, greedy=False):
        trainExamples = []
        board = self.game.getInitBoard()
        items_list_board = self.game.getInitItems(self.items_list)
        while True:
            state = self.game.getBinItem(board, items_list_board)
            pi = self.mcts.getActionProb(state, self.items_total_area, self.rewards_list, greedy_a=0 if greedy else 1)
            trainExamples.append([state, pi, None])
            np.random.seed()
            action = np.random.choice(len(pi), p=pi)
            board, items_list_board = self.game.getNextState(board, action, items_list_board)
            r, score = self.game.getGameEnded(self.game.getBinItem(board, items_list_board), self.items_total_area,
                                              self.rewards_list, self.args.alpha)
            if r != 0:
                self.ep_score = score
                return [(x[0], x[1], r) for x in trainExamples]

    # ---- batched iteration ------------------------------------------------------------------------------------------
    def _sample_seed(self):
        """Seed of the move-sampling streams, the same on every rank: args.sample_seed if given, else OS entropy drawn on rank 0
        (the reference reseeds from OS entropy before every np.random.choice, CoachBPP.py:86-87)."""
        import torch
        s = _opt(self.args, "sample_seed", None)
        if s is None:
            s = int.from_bytes(os.urandom(7), "little")
            if rdist.collectives_on():
                t = torch.tensor([s], dtype=torch.int64, device=self.nnet.device)
                torch.distributed.broadcast(t, src=0)
                s = int(t.item())
        return int(s) & 0x7FFFFFFFFFFFFFFF

    def _driver(self, n_eps):
        """The batched self-play driver, sized for `n_eps` episodes per iteration; rebuilt when a later iteration asks for more."""
        world = rdist.world_size()
        per_rank = max(1, (int(n_eps) + world - 1) // world)
        if self._selfplay is not None and per_rank > self._selfplay_per_rank:
            self._selfplay.close()
            self._selfplay = None
        if self._selfplay is None:
            from .selfplay import BatchedSelfPlay
            if not hasattr(self, "_seed"):
                self._seed = self._sample_seed()
            games = min(int(_opt(self.args, "games_per_gpu", 0) or per_rank), per_rank)
            moves_cap = self.game.num_items
            self._selfplay = BatchedSelfPlay(self.game, self.nnet, self.args, games=games, move_rule=_lib.MOVE_SAMPLE,
                                             seed=self._seed,  # the same on every rank
                                             node_cap=int(_opt(self.args, "node_cap", 0) or 0),
                                             edge_cap=int(_opt(self.args, "edge_cap", 0) or 0), vis_cap=int(_opt(self.args, "vis_cap", 0) or 0),
                                             max_examples=per_rank * moves_cap + 64, use_graph=bool(_opt(self.args, "use_graph", True)),
                                             groups=max(1, min(int(_opt(self.args, "groups", 2) or 2), games)),
                                             tie_salt=_opt(self.args, "tie_salt", None), host_evaluator=_opt(self.args, "host_evaluator", None))
            self._selfplay_per_rank = per_rank
            self._move_mode = None
        return self._selfplay

    def drawIteration(self):
        """(generator height, [generator seed per episode]) of one iteration (CoachBPP.py:117-118,127), identical on every rank.
        The reference draws them from an OS-seeded stream one episode at a time; here they are drawn up front."""
        import torch
        args = self.args
        np.random.seed()  # CoachBPP.py:117
        draw = [np.random.randint(args.binH_min, args.binH + 1)] + [np.random.randint(int(1e5)) for _ in range(args.numEps)]
        if rdist.collectives_on():  # every rank must play the same instances
            t = torch.tensor(draw, dtype=torch.int64, device=self.nnet.device)
            torch.distributed.broadcast(t, src=0)
            draw = t.cpu().tolist()
        return int(draw[0]), [int(x) for x in draw[1:]]

    def selfPlayIteration(self, i, draws=None, move_rule=None):
        """Self-play of iteration i: returns (ep_scores in episode order, the PackedReplay of all ranks' examples in the reference's
        order: episode by episode, move by move).  draws: (generator height, seeds) instead of drawIteration()'s (tests pin the
        reference's captured draws); move_rule: overrides sampling / greedy (tests: argmax moves with proportional targets)."""
        import time
        import torch
        args = self.args
        bin_height, seeds = self.drawIteration() if draws is None else (int(draws[0]), [int(x) for x in draws[1]])
        n_eps = len(seeds)
        self.gen.bin_height = bin_height  # :118
        self.items_total_area = self.gen.bin_height * self.gen.bin_width  # :119
        mine = rdist.shard(n_eps)
        sp = self._driver(n_eps)
        greedy = i > args.iterStepThreshold  # :132
        mode = (_lib.MOVE_ARGMAX_FIRST if greedy else _lib.MOVE_SAMPLE, bool(greedy)) if move_rule is None else (int(move_rule), bool(greedy))
        if getattr(self, "_move_mode", None) != mode:
            sp.set_move_rule(mode[0], onehot_examples=mode[1])  # re-captures the waves
            self._move_mode = mode
        sp.clear_examples()
        # global episode ids: base + index, base = the episodes this Coach has played so far -- the sampling stream of (episode, move)
        # is never reused by a later iteration, and does not depend on which rank or slot plays the episode
        base = self.episodes_played
        self.episodes_played += n_eps
        dev = self.nnet.device
        t0 = time.time()
        if mine:
            first = base + mine[0]
            seeds_mine = np.asarray([seeds[k] for k in mine], dtype=np.uint32)
            from .binpacking.BinPackingGame import ItemsGenerator
            if _opt(args, "host_items", False) or not isinstance(self.gen, ItemsGenerator):  # instances through gen.items_generator on the host (:127-130)
                state = np.random.get_state()  # items_generator reseeds the global stream (BinPackingGame.py:258)
                wh = np.array([[it[:2] for it in self.gen.items_generator(int(sd))] for sd in seeds_mine], dtype=np.uint8)
                np.random.set_state(state)
                ids, outcome, score, moves, stats = sp.run(wh, np.full(len(mine), self.items_total_area, np.int32), self.rewards_list, first_id=first)
            else:  # bit-identical instances generated on the device (k_items_generator), total area W * bin_height
                ids, outcome, score, moves, stats = sp.run_from_seeds(seeds_mine, self.rewards_list, first_id=first, bin_h=bin_height, bin_w=self.gen.bin_width)
            local = torch.zeros(len(mine), dtype=torch.float64, device=dev)
            local[torch.as_tensor(ids.astype(np.int64) - first, device=dev)] = torch.as_tensor(score, device=dev)
            replay = sp.examples_packed()
        else:  # more ranks than episodes: nothing to play, but every collective below is still joined
            from .replay import PackedReplay
            stats = {}
            local = torch.zeros(0, dtype=torch.float64, device=dev)
            replay = PackedReplay.empty(self.game.bin_width, self.game.bin_height, self.game.num_items, sp.eng.KW, dev)
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        t_play = time.time() - t0
        self.last_example_keys = replay.episode * (self.game.num_items + 1) + replay.move.to(torch.int64) - base * (self.game.num_items + 1)  # LOCAL examples (tests)
        if rdist.collectives_on():
            mine_t = torch.as_tensor(mine, dtype=torch.int64, device=dev)
            gathered = rdist.all_gather_variable(torch.stack([mine_t.to(torch.float64), local], dim=1))
            ep_scores = np.zeros(n_eps)
            g = gathered.cpu().numpy()
            ep_scores[g[:, 0].astype(np.int64)] = g[:, 1]
            replay = rdist.all_gather_packed(replay)  # rank order -> (episode, move) order, the same on every rank
            exch = dict(rdist.last_exchange)
        else:
            ep_scores = local.cpu().numpy()
            exch = dict(bytes_sent=0, bytes_received=0, ms=0.0, examples=len(replay))
        self.last_stats = stats
        self.timings.append(dict(iteration=i, episodes=n_eps, selfplay_s=t_play, examples=len(replay), replay_bytes=replay.nbytes, exchange=exch))
        return [float(s) for s in ep_scores], replay

    def learn(self):
        import time
        import torch
        from .replay import PackedReplay
        args = self.args
        if rdist.collectives_on() and getattr(self.nnet, "grad_hook", None) is None:
            rdist.attach(self.nnet)  # identical weights on every rank, gradients summed over the ranks' batch slices
        for i in range(1, args.numIters + 1):
            log.info("Starting Iter #%d ...", i)
            if not self.skipFirstSelfPlay or i > 1:
                ep_scores, examples = self.selfPlayIteration(i)
                self.iteration_scores.append(list(ep_scores))
                self.rewards_list.extend(ep_scores)  # :134, in episode order
                self.rewards_list = trim_min(self.rewards_list, int(args.numScoresForRank))  # :136-139 drop the smallest scores
                self.ep_score = ep_scores[-1]
                metrics = {"iter mean reward": float(np.mean(ep_scores)),
                           "optimality percentage": sum(s == 1.0 for s in ep_scores) / len(ep_scores),
                           "min reward": float(np.min(ep_scores)), "max reward": float(np.max(ep_scores))}  # :143-147
                self.metrics_log.append(dict(metrics, iteration=i))
                log.info("iter %d: %s", i, metrics)
                # deque(maxlen=maxlenOfQueue) (:122): the LAST maxlen examples in episode order
                self.trainExamplesHistory.append(examples.tail(int(args.maxlenOfQueue)))
            if len(self.trainExamplesHistory) > args.numItersForTrainExamplesHistory:  # :154-157
                log.warning("Removing the oldest entry in trainExamples. len(trainExamplesHistory) = %d", len(self.trainExamplesHistory))
                self.trainExamplesHistory.pop(0)
            train_set = PackedReplay.cat(self.trainExamplesHistory)
            if rdist.rank() == 0:
                self.nnet.save_checkpoint(folder=args.checkpoint, filename="temp.pth.tar")  # :172
            t0 = time.time()
            self.nnet.train_packed(train_set)  # :176 (sampling is with replacement, so no shuffle is needed)
            if self.nnet.device.type == "cuda":
                torch.cuda.synchronize(self.nnet.device)
            if self.timings:
                self.timings[-1].update(train_s=time.time() - t0, train_steps=int(getattr(self.nnet, "last_train_steps", 0)), train_examples=len(train_set),
                                        train_set_bytes=train_set.nbytes)
            if rdist.rank() == 0:
                self.save_rewards_list()  # :196

    # ---- files (formats of CoachBPP.py:198-231) ----------------------------------------------------------------------
    def save_rewards_list(self):
        os.makedirs(self.args.checkpoint, exist_ok=True)
        path = os.path.join(self.args.checkpoint, "rewards_list_" + str(self.args.numItems) + "_items.pkl")
        with open(path, "wb") as f:
            pickle.dump([float(x) for x in self.rewards_list], f)

    def getCheckpointFile(self, iteration):
        return "checkpoint_" + ".pth.tar"

    def saveTrainExamples(self, iteration):
        import torch
        os.makedirs(self.args.checkpoint, exist_ok=True)
        path = os.path.join(self.args.checkpoint, self.getCheckpointFile(iteration) + ".examples")
        torch.save([dict(W=e.W, H=e.H, N=e.N, flat=e.to_flat().cpu()) for e in self.trainExamplesHistory], path)

    def loadTrainExamples(self):
        import torch
        path = os.path.join(self.args.load_folder_file[0], self.args.load_folder_file[1]) + ".examples"
        if not os.path.isfile(path):
            raise FileNotFoundError('File "%s" with trainExamples not found' % path)  # the reference prompts on stdin here
        hist = torch.load(path, weights_only=True)
        from .replay import PackedReplay
        self.trainExamplesHistory = [PackedReplay.from_flat(e["flat"].to(self.nnet.device), e["W"], e["H"], e["N"]) for e in hist]
        self.skipFirstSelfPlay = True
