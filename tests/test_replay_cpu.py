"""Host logic of the packed replay set (resource_packing_self_play_amd/replay.py) on CPU tensors: concatenation shifts pool offsets,
select / tail share then compact the pool, the flat byte form round-trips, (episode, move) ordering; and the two-rank exchange over
gloo.  The expansion to planes / pi runs on the GPU only (tests/test_gpu_coach.py)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from resource_packing_self_play_amd.replay import PackedReplay

W, H, N, KW = 10, 10, 8, 11


def synthetic(n, seed, episode0=0):
    g = np.random.default_rng(seed)
    sp_n = g.integers(1, 6, size=n).astype(np.int32)
    off = np.cumsum(sp_n) - sp_n
    S = int(sp_n.sum())
    ep = episode0 + np.sort(g.integers(0, 3, size=n)).astype(np.int64)
    mv = np.zeros(n, np.int32)
    for e in np.unique(ep):
        sel = np.nonzero(ep == e)[0]
        mv[sel] = g.permutation(len(sel))
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a)).to(dt)
    return PackedReplay(W, H, N, t(g.integers(0, 1 << 20, size=(n, KW)), torch.int32), t(g.integers(1, 10, size=(n, 2 * N)), torch.uint8),
                        t(g.choice([-1, 1], size=n), torch.int32), t(off, torch.int64), t(sp_n, torch.int32),
                        t(g.integers(0, W * N, size=S), torch.int16), t(g.integers(1, 400, size=S), torch.int32), t(ep, torch.int64), t(mv, torch.int32))


def pairs(r, k):
    o, n = int(r.sp_off[k]), int(r.sp_n[k])
    return r.sp_act[o:o + n].tolist(), r.sp_cnt[o:o + n].tolist()


def same_examples(a, ia, b, ib):
    return (torch.equal(a.key[ia], b.key[ib]) and torch.equal(a.wh[ia], b.wh[ib]) and int(a.value[ia]) == int(b.value[ib]) and pairs(a, ia) == pairs(b, ib))


def test_cat_select_tail_compact_and_flat_round_trip():
    a, b = synthetic(7, 1), synthetic(5, 2, episode0=10)
    c = PackedReplay.cat([a, b])
    assert len(c) == 12 and c.sp_act.shape[0] == a.sp_act.shape[0] + b.sp_act.shape[0]
    assert all(same_examples(c, k, a, k) for k in range(7)) and all(same_examples(c, 7 + k, b, k) for k in range(5))
    order = torch.tensor([11, 0, 5, 5, 3])
    s = c.select(order)
    assert s.sp_act.data_ptr() == c.sp_act.data_ptr() and all(same_examples(s, k, c, int(order[k])) for k in range(5))
    t = c.tail(4)
    assert len(t) == 4 and t.sp_act.shape[0] == int(t.sp_n.sum()) and all(same_examples(t, k, c, 8 + k) for k in range(4))
    assert c.tail(100) is c and len(c.tail(0)) == 0
    srt = c.sort_by_episode_move()
    k = (srt.episode * (N + 1) + srt.move).tolist()
    assert k == sorted(k) and len(srt) == 12
    flat = s.to_flat()
    assert flat.dtype == torch.uint8 and flat.numel() % 8 == 0
    back = PackedReplay.from_flat(flat, W, H, N)
    assert len(back) == 5 and back.sp_act.shape[0] == int(s.sp_n.sum()) and all(same_examples(back, k, s, k) for k in range(5))
    assert torch.equal(back.episode, s.episode) and torch.equal(back.move, s.move)
    e = PackedReplay.from_flat(PackedReplay.empty(W, H, N, KW, torch.device("cpu")).to_flat(), W, H, N)
    assert len(e) == 0 and len(PackedReplay.cat([e, a])) == 7
    # the size claim of DESIGN.md: a 20x20 / 32-item example with 40 visited root edges is under 0.5 KB
    assert 4 * (20 + 1) + 64 + 4 + 8 + 4 + 40 * 6 < 512


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from resource_packing_self_play_amd import distributed as rdist
    rdist.init_from_env(backend="gloo")
    mine = synthetic(4 + 3 * rank, 100 + rank, episode0=5 * (1 - rank))  # rank 1 holds the EARLIER episodes: the union must be re-ordered
    got = rdist.all_gather_packed(mine)
    parts = [synthetic(4 + 3 * r, 100 + r, episode0=5 * (1 - r)) for r in range(world)]
    want = PackedReplay.cat(parts).sort_by_episode_move()
    assert len(got) == len(want) == 11 and all(same_examples(got, k, want, k) for k in range(11))
    assert torch.equal(got.episode, want.episode) and torch.equal(got.move, want.move) and int(got.episode[0]) == 0
    ex = rdist.last_exchange
    assert ex["examples"] == 11 and ex["bytes_sent"] == mine.to_flat().numel() and ex["bytes_received"] == sum(p.to_flat().numel() for p in parts)
    empty = PackedReplay.empty(W, H, N, KW, torch.device("cpu")) if rank == 1 else mine  # a rank without episodes still joins
    got = rdist.all_gather_packed(empty)
    assert len(got) == 4
    dist.barrier()
    open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    dist.destroy_process_group()


def test_two_ranks_exchange_packed_replay_over_gloo(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(os.path.join(str(tmp_path), "ok0")) and os.path.exists(os.path.join(str(tmp_path), "ok1"))
