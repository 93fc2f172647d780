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
    (CoachBPP.py:86-87), greedy ties go to the lowest action instead of a random one (MCTS_bpp.py:45-46);
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
        self.trainExamplesHistory = []  # one (planes, pi, value) tensor triple per iteration (device tensors)
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
    def _driver(self, n_eps):
        if self._selfplay is None:
            from .selfplay import BatchedSelfPlay
            world = rdist.world_size()
            per_rank = (int(n_eps) + world - 1) // world
            games = int(getattr(self.args, "games_per_gpu", 0) or per_rank)
            moves_cap = self.game.num_items
            self._selfplay = BatchedSelfPlay(self.game, self.nnet, self.args, games=min(games, per_rank), move_rule=_lib.MOVE_SAMPLE,
                                             seed=int(getattr(self.args, "seed", 0) or 0) * 1000003 + 17,  # the same on every rank
                                             node_cap=int(getattr(self.args, "node_cap", 0) or 0),
                                             edge_cap=int(getattr(self.args, "edge_cap", 0) or 0),
                                             max_examples=2 * per_rank * moves_cap, use_graph=bool(getattr(self.args, "use_graph", True)),
                                             groups=int(getattr(self.args, "groups", 2) or 2),
                                             tie_salt=getattr(self.args, "tie_salt", None), host_evaluator=getattr(self.args, "host_evaluator", None))
        return self._selfplay

    def drawIteration(self):
        """(generator height, [generator seed per episode]) of one iteration (CoachBPP.py:117-118,127), identical on every rank.
        The reference draws them from an OS-seeded stream one episode at a time; here they are drawn up front."""
        import torch
        args = self.args
        np.random.seed()  # CoachBPP.py:117
        draw = [np.random.randint(args.binH_min, args.binH + 1)] + [np.random.randint(int(1e5)) for _ in range(args.numEps)]
        if rdist.world_size() > 1:  # every rank must play the same instances
            t = torch.tensor(draw, dtype=torch.int64, device=self.nnet.device)
            torch.distributed.broadcast(t, src=0)
            draw = t.cpu().tolist()
        return int(draw[0]), [int(x) for x in draw[1:]]

    def selfPlayIteration(self, i, draws=None, move_rule=None):
        """Self-play of iteration i: returns (ep_scores in episode order, (planes, pi, value) of all ranks in the reference's order:
        episode by episode, move by move).  draws: (generator height, seeds) instead of drawIteration()'s (tests pin the
        reference's captured draws); move_rule: overrides sampling / greedy (tests: argmax moves with proportional targets)."""
        import torch
        args = self.args
        bin_height, seeds = self.drawIteration() if draws is None else (int(draws[0]), [int(x) for x in draws[1]])
        n_eps = len(seeds)
        self.gen.bin_height = bin_height  # :118
        self.items_total_area = self.gen.bin_height * self.gen.bin_width  # :119
        state = np.random.get_state()  # items_generator reseeds the global stream (BinPackingGame.py:258)
        wh = np.array([[it[:2] for it in self.gen.items_generator(s)] for s in seeds], dtype=np.uint8)  # :127-130
        np.random.set_state(state)
        mine = rdist.shard(n_eps)
        sp = self._driver(n_eps)
        greedy = i > args.iterStepThreshold  # :132
        mode = (_lib.MOVE_ARGMAX_FIRST if greedy else _lib.MOVE_SAMPLE, bool(greedy)) if move_rule is None else (int(move_rule), bool(greedy))
        if getattr(self, "_move_mode", None) != mode:
            sp.set_move_rule(mode[0], onehot_examples=mode[1])  # re-captures the waves
            self._move_mode = mode
        sp.clear_examples()
        first = mine[0] if mine else 0  # episode ids are global: first .. first + len(mine) - 1
        ids, outcome, score, moves, stats = sp.run(wh[mine], np.full(len(mine), self.items_total_area, np.int32), self.rewards_list,
                                                   first_id=first)
        dev = self.nnet.device
        mine_t = torch.as_tensor(mine, dtype=torch.int64, device=dev)
        local = torch.zeros(len(mine), dtype=torch.float64, device=dev)
        local[torch.as_tensor(ids.astype(np.int64) - first, device=dev)] = torch.as_tensor(score, device=dev)
        planes, pi, value, ex_ep, ex_mv = sp.examples(with_meta=True)
        key = torch.as_tensor(ex_ep.astype(np.int64) * (self.game.num_items + 1) + ex_mv.astype(np.int64), device=dev)
        if rdist.world_size() > 1:
            gathered = rdist.all_gather_variable(torch.stack([mine_t.to(torch.float64), local], dim=1))
            ep_scores = np.zeros(n_eps)
            g = gathered.cpu().numpy()
            ep_scores[g[:, 0].astype(np.int64)] = g[:, 1]
            planes, pi, value = rdist.all_gather_examples(planes, pi, value)
            order = torch.argsort(rdist.all_gather_variable(key))  # rank order -> (episode, move) order, the same on every rank
            planes, pi, value = planes.index_select(0, order), pi.index_select(0, order), value.index_select(0, order)
        else:
            ep_scores = local.cpu().numpy()
        self.last_stats = stats
        self.last_example_keys = key  # episode * (N + 1) + move of every LOCAL example (tests)
        return [float(s) for s in ep_scores], (planes, pi, value)

    def learn(self):
        import torch
        args = self.args
        if rdist.world_size() > 1 and getattr(self.nnet, "grad_hook", None) is None:
            rdist.attach(self.nnet)  # identical weights on every rank, gradients summed over the ranks' batch slices
        for i in range(1, args.numIters + 1):
            log.info("Starting Iter #%d ...", i)
            if not self.skipFirstSelfPlay or i > 1:
                ep_scores, examples = self.selfPlayIteration(i)
                self.iteration_scores.append(list(ep_scores))
                self.rewards_list.extend(ep_scores)  # :134, in episode order
                while len(self.rewards_list) > args.numScoresForRank:  # :136-139 drop the smallest score
                    self.rewards_list.pop(int(np.argmin(self.rewards_list)))
                self.ep_score = ep_scores[-1]
                metrics = {"iter mean reward": float(np.mean(ep_scores)),
                           "optimality percentage": sum(s == 1.0 for s in ep_scores) / len(ep_scores),
                           "min reward": float(np.min(ep_scores)), "max reward": float(np.max(ep_scores))}  # :143-147
                self.metrics_log.append(dict(metrics, iteration=i))
                log.info("iter %d: %s", i, metrics)
                keep = int(args.maxlenOfQueue)  # deque(maxlen=maxlenOfQueue) (:122): the LAST maxlen examples in episode order
                self.trainExamplesHistory.append(tuple(t[-keep:] for t in examples))
            if len(self.trainExamplesHistory) > args.numItersForTrainExamplesHistory:  # :154-157
                log.warning("Removing the oldest entry in trainExamples. len(trainExamplesHistory) = %d", len(self.trainExamplesHistory))
                self.trainExamplesHistory.pop(0)
            planes = torch.cat([e[0] for e in self.trainExamplesHistory])
            pi = torch.cat([e[1] for e in self.trainExamplesHistory])
            value = torch.cat([e[2] for e in self.trainExamplesHistory])
            if rdist.rank() == 0:
                self.nnet.save_checkpoint(folder=args.checkpoint, filename="temp.pth.tar")  # :172
            self.nnet.train_tensors(planes, pi, value)  # :176 (sampling is with replacement, so no shuffle is needed)
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
        torch.save([tuple(t.cpu() for t in e) for e in self.trainExamplesHistory], path)

    def loadTrainExamples(self):
        import torch
        path = os.path.join(self.args.load_folder_file[0], self.args.load_folder_file[1]) + ".examples"
        if not os.path.isfile(path):
            raise FileNotFoundError('File "%s" with trainExamples not found' % path)  # the reference prompts on stdin here
        hist = torch.load(path, weights_only=True)
        self.trainExamplesHistory = [tuple(t.to(self.nnet.device) for t in e) for e in hist]
        self.skipFirstSelfPlay = True
