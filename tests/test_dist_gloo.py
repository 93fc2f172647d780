"""The multi-GPU plumbing (resource_packing_self_play_amd.distributed) on CPU: world_size 2, gloo backend."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from resource_packing_self_play_amd import distributed as rdist
    r, w, _ = rdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and rdist.rank() == rank and rdist.world_size() == world
    # episode sharding is a partition that does not depend on slot counts
    assert rdist.shard(7) == list(range(rank, 7, world))
    # variable-length all-gather keeps rank order
    t = torch.arange(3 + 2 * rank, dtype=torch.float32).reshape(-1, 1) + 100 * rank
    g = rdist.all_gather_variable(t)
    want = torch.cat([torch.arange(3 + 2 * k, dtype=torch.float32).reshape(-1, 1) + 100 * k for k in range(world)])
    assert torch.equal(g, want)
    planes = (torch.rand(2 + rank, 3, 4, 4) < 0.5).float(); pi = torch.rand(2 + rank, 12); val = torch.ones(2 + rank) * (rank + 1)
    P, Pi, V = rdist.all_gather_examples(planes, pi, val)
    assert P.shape[0] == sum(2 + k for k in range(world)) and P.dtype == torch.float32
    assert torch.equal(P[sum(2 + k for k in range(rank)):][:2 + rank], planes) and V.tolist() == sum([[k + 1.0] * (2 + k) for k in range(world)], [])
    # data-parallel step: averaged gradients of two half batches == gradient of the full batch
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 2))
    class Wrap: pass
    wr = Wrap(); wr.nnet = model; wr.grad_hook = None
    if rank == 1:
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)  # diverge on purpose; attach() must broadcast rank 0's weights
    rdist.attach(wr)
    x = torch.arange(48, dtype=torch.float32).reshape(8, 6) / 10; y = torch.arange(16, dtype=torch.float32).reshape(8, 2) / 7
    half = slice(4 * rank, 4 * rank + 4)
    loss = ((model(x[half]) - y[half]) ** 2).sum() / 4
    loss.backward()
    wr.grad_hook(model)
    ref = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 2))
    torch.manual_seed(0)
    ref = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 2))
    (((ref(x) - y) ** 2).sum() / 8).backward()
    for p, q in zip(model.parameters(), ref.parameters()):
        assert torch.allclose(p.grad, q.grad, atol=1e-6)
    dist.barrier()
    open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    dist.destroy_process_group()


def test_world_size_2_gloo(tmp_path):
    port = free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(os.path.join(str(tmp_path), "ok0")) and os.path.exists(os.path.join(str(tmp_path), "ok1"))


def test_single_process_is_a_no_op():
    from resource_packing_self_play_amd import distributed as rdist
    t = torch.arange(5.0)
    assert rdist.world_size() == 1 and rdist.rank() == 0 and rdist.shard(5) == [0, 1, 2, 3, 4]
    assert rdist.all_gather_variable(t) is t
