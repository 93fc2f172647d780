"""Multi-GPU plumbing: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on ROCm, "gloo" on CPU
for tests).  Self-play episodes shard across ranks with no data-path collective -- each rank owns whole games
(CoachBPP.py:123-134 runs them independently).  Only two exchanges exist, once per CoachBPP iteration:

  * all-gather of the replay examples and episode scores (variable length: sizes first, then padded payloads), so every
    rank rebuilds the same training set and the same R2 buffer;
  * all-reduce (sum) of the FP32 gradients as ONE flat bucket per optimizer step -- 0.6-8.7 MB, latency-bound on
    7 x 153 GB/s xGMI links, so a single message beats per-tensor calls.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* as torchrun sets them.  Returns (rank, world, local device index).
    Rehearsal on a single-GPU box: RP_DIST_BACKEND=gloo RP_SINGLE_DEVICE=1 runs every rank on device 0 with gloo."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = 0 if os.environ.get("RP_SINGLE_DEVICE") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("RP_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_initialized() else 0


def shard(n_items, rank_=None, world=None):
    """Indices of the episodes rank `rank_` plays: i = rank, rank + world, ...  Independent of how many slots a rank
    has, so results do not depend on the rank count."""
    rank_ = rank() if rank_ is None else rank_
    world = world_size() if world is None else world
    return list(range(rank_, n_items, world))


def all_gather_variable(t):
    """All-gathers tensors that differ in their first dimension; returns the concatenation in rank order."""
    if world_size() == 1:
        return t
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(world_size())]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    cap = max(sizes)
    pad = torch.zeros((cap,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[:t.shape[0]] = t
    out = [torch.empty_like(pad) for _ in range(world_size())]
    dist.all_gather(out, pad)
    return torch.cat([o[:s] for o, s in zip(out, sizes)], dim=0)


def all_gather_examples(planes, pi, value):
    """Replay exchange of one iteration.  Planes travel as uint8 (they are 0/1), 4x fewer bytes than FP32."""
    if world_size() == 1:
        return planes, pi, value
    p8 = all_gather_variable(planes.to(torch.uint8))
    return p8.to(torch.float32), all_gather_variable(pi), all_gather_variable(value)


class FlatGradAllReduce:
    """Gradient hook for NNetWrapper.train_tensors: one flat FP32 bucket, one all-reduce(sum), mean over ranks."""

    def __init__(self, module):
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        self.flat = None

    def __call__(self, module):
        if world_size() == 1:
            return
        dev = self.params[0].device
        if self.flat is None or self.flat.device != dev:
            self.flat = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                self.flat[off:off + n].zero_()
            else:
                self.flat[off:off + n].copy_(p.grad.reshape(-1))
            off += n
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        self.flat.div_(world_size())
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                p.grad = torch.empty_like(p)
            p.grad.copy_(self.flat[off:off + n].view_as(p))
            off += n


def broadcast_parameters(module, src=0):
    if world_size() == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src)


def attach(nnet_wrapper):
    """Makes `nnet_wrapper.train*` data-parallel: identical initial weights, averaged gradients every step."""
    broadcast_parameters(nnet_wrapper.nnet)
    nnet_wrapper.grad_hook = FlatGradAllReduce(nnet_wrapper.nnet)
    return nnet_wrapper
