"""GPU probe: is the self-play loop bound by the host's graph launches?  Plays the bench workload for a while, then times
(a) the host side of `step()` alone (no sync inside) and (b) the same number of waves end to end, for 1..4 slot groups."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from resource_packing_self_play_amd import _lib
from resource_packing_self_play_amd.binpacking.BinPackingGame import BinPackingGame
from resource_packing_self_play_amd.binpacking.pytorch.NNet import NNetWrapper
from resource_packing_self_play_amd.selfplay import BatchedSelfPlay
from resource_packing_self_play_amd.utils import dotdict
import bench

W = H = 20; N = 32; sims = 24
games = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
for groups in [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "3").split(",")]:
    game = BinPackingGame(W, H, N, 1)
    args = dotdict(numMCTSSims=sims, cpuct=1, alpha=0.75, cuda=True, num_items=N, num_bins=1, epochs=1, batch_size=64)
    torch.manual_seed(0)
    nnet = NNetWrapper(game, args)
    node_cap = sims * (N + 1) + 2
    sp = BatchedSelfPlay(game, nnet, args, games=games, move_rule=_lib.MOVE_SAMPLE, seed=7, node_cap=node_cap, edge_cap=30 * 4096,
                         vis_cap=30 * 1024, groups=groups, reclaim=True)
    sp.prepare()
    wh = bench.make_instances(W, H, N, games, 100)
    sp.start(wh, np.full(games, W * H, np.int32), bench.rank_buffer(), first_id=0)
    for _ in range(30):
        sp.step()
    torch.cuda.synchronize()
    n = 100
    t0 = time.perf_counter()
    for _ in range(n):
        sp.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("groups %d x %d slots: host %.3f ms per step() (%.3f per graph launch), end to end %.3f ms per step (%.3f per wave)" %
          (groups, games // groups, (t1 - t0) / n * 1e3, (t1 - t0) / n / groups * 1e3, (t2 - t0) / n * 1e3, (t2 - t0) / n / groups * 1e3), flush=True)
    # raw launch cost of one graph when the stream is idle
    g = sp.groups[0]
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        t0 = time.perf_counter(); sp.step_group(g); ts.append(time.perf_counter() - t0); torch.cuda.synchronize()
    print("   idle-stream graph launch: host %.3f ms (min %.3f)" % (np.mean(ts) * 1e3, np.min(ts) * 1e3), flush=True)
    del sp, nnet
    torch.cuda.empty_cache()
