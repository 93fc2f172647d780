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
  * the per-episode generator seeds are drawn up front from one OS-seeded stream.
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
    def _driver(self):
        if self._selfplay is None:
            from .selfplay import BatchedSelfPlay
            world = rdist.world_size()
            per_rank = (int(self.args.numEps) + world - 1) // world
            games = int(getattr(self.args, "games_per_gpu", 0) or per_rank)
            moves_cap = self.game.num_items
            self._selfplay = BatchedSelfPlay(self.game, self.nnet, self.args, games=min(games, per_rank), move_rule=_lib.MOVE_SAMPLE,
                                             seed=int(getattr(self.args, "seed", 0) or 0) * 1000003 + rdist.rank(),
                                             node_cap=int(getattr(self.args, "node_cap", 0) or 0),
                                             edge_cap=int(getattr(self.args, "edge_cap", 0) or 0),
                                             max_examples=2 * per_rank * moves_cap, use_graph=bool(getattr(self.args, "use_graph", True)),
                                             groups=int(getattr(self.args, "groups", 2) or 2))
        return self._selfplay

    def selfPlayIteration(self, i):
        """Self-play of iteration i: returns (ep_scores in episode order, (planes, pi, value) of all ranks)."""
        import torch
        args = self.args
        np.random.seed()  # CoachBPP.py:117
        if rdist.world_size() > 1:  # every rank must draw the same instances
            draw = torch.tensor([np.random.randint(args.binH_min, args.binH + 1)] + [np.random.randint(int(1e5)) for _ in range(args.numEps)],
                                dtype=torch.int64, device=self.nnet.device)
            torch.distributed.broadcast(draw, src=0)
            draw = draw.cpu().tolist()
        else:
            draw = [np.random.randint(args.binH_min, args.binH + 1)] + [np.random.randint(int(1e5)) for _ in range(args.numEps)]
        self.gen.bin_height = int(draw[0])  # :118
        self.items_total_area = self.gen.bin_height * self.gen.bin_width  # :119
        seeds = draw[1:]
        wh = np.array([[it[:2] for it in self.gen.items_generator(s)] for s in seeds], dtype=np.uint8)  # :127-130
        mine = rdist.shard(args.numEps)
        sp = self._driver()
        greedy = i > args.iterStepThreshold  # :132
        if getattr(self, "_greedy_mode", None) != greedy:
            sp.set_move_rule(_lib.MOVE_ARGMAX_FIRST if greedy else _lib.MOVE_SAMPLE, onehot_examples=greedy)  # re-captures the waves
            self._greedy_mode = greedy
        sp.clear_examples()
        ids, outcome, score, moves, stats = sp.run(wh[mine], np.full(len(mine), self.items_total_area, np.int32), self.rewards_list,
                                                   first_id=0)
        local = torch.zeros(len(mine), dtype=torch.float64, device=self.nnet.device)
        local[torch.as_tensor(ids.astype(np.int64), device=self.nnet.device)] = torch.as_tensor(score, device=self.nnet.device)
        if rdist.world_size() > 1:
            gathered = rdist.all_gather_variable(torch.stack([torch.as_tensor(mine, dtype=torch.float64, device=self.nnet.device), local], dim=1))
            ep_scores = np.zeros(args.numEps)
            g = gathered.cpu().numpy()
            ep_scores[g[:, 0].astype(np.int64)] = g[:, 1]
        else:
            ep_scores = local.cpu().numpy()
        examples = rdist.all_gather_examples(*sp.examples())
        self.last_stats = stats
        return [float(s) for s in ep_scores], examples

    def learn(self):
        import torch
        args = self.args
        for i in range(1, args.numIters + 1):
            log.info("Starting Iter #%d ...", i)
            if not self.skipFirstSelfPlay or i > 1:
                ep_scores, examples = self.selfPlayIteration(i)
                self.rewards_list.extend(ep_scores)  # :134, in episode order
                while len(self.rewards_list) > args.numScoresForRank:  # :136-139 drop the smallest score
                    self.rewards_list.pop(int(np.argmin(self.rewards_list)))
                self.ep_score = ep_scores[-1]
                metrics = {"iter mean reward": float(np.mean(ep_scores)),
                           "optimality percentage": sum(s == 1.0 for s in ep_scores) / len(ep_scores),
                           "min reward": float(np.min(ep_scores)), "max reward": float(np.max(ep_scores))}  # :143-147
                self.metrics_log.append(dict(metrics, iteration=i))
                log.info("iter %d: %s", i, metrics)
                keep = int(args.maxlenOfQueue)  # deque(maxlen=maxlenOfQueue) (:122)
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
