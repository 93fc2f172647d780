"""Multi-GPU plumbing: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on ROCm, "gloo" on CPU
for tests).  Self-play episodes shard across ranks with no data-path collective -- each rank owns whole games
(CoachBPP.py:123-134 runs them independently).  Only two exchanges exist, once per CoachBPP iteration:

  * all-gather of the replay examples and episode scores (variable length: sizes first, then padded payloads), so every
    rank rebuilds the same training set and the same R2 buffer;
  * all-reduce (sum) of the FP32 gradients as ONE flat bucket per optimizer step -- 0.6-8.7 MB, latency-bound on
    7 x 153 GB/s xGMI links, so a single message beats per-tensor calls.  A step covers `batch_size` examples whatever the
    rank count: one shared index stream, rank r takes every world-th index (NNetWrapper.train_tensors).
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
    if world > 1:
        # one MIOpen user database / kernel cache per rank: the ranks of a node meet every new convolution shape (training batches,
        # first waves) at the same moment, and concurrent writers of one sqlite file are a known source of rare start-up failures
        base = os.environ.get("RP_MIOPEN_DIR", "/tmp/rp_miopen_%d" % os.getuid())
        os.environ.setdefault("MIOPEN_USER_DB_PATH", os.path.join(base, "rank%d" % rank, "db"))
        os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", os.path.join(base, "rank%d" % rank, "cache"))
        for k in ("MIOPEN_USER_DB_PATH", "MIOPEN_CUSTOM_CACHE_DIR"):
            os.makedirs(os.environ[k], exist_ok=True)
    # RP_DIST_FORCE=1: form the process group even for one rank (RCCL smoke run on a single-GPU box: backend initialisation and the
    # device-tensor collectives are the same code at any world size)
    if (world > 1 or os.environ.get("RP_DIST_FORCE") == "1") and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("RP_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def collectives_on():
    """Collectives run when there is more than one rank -- or when a one-rank group was formed on purpose (RP_DIST_FORCE)."""
    return dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("RP_DIST_FORCE") == "1")


def rank():
    return dist.get_rank() if dist.is_initialized() else 0


def shard(n_items, rank_=None, world=None):
    """Episodes rank `rank_` plays: a contiguous block of the iteration's episodes (sizes differ by at most one).  An episode's
    result depends on its global index only (instance, sampling stream), never on the rank or slot that played it."""
    rank_ = rank() if rank_ is None else rank_
    world = world_size() if world is None else world
    base, rem = divmod(int(n_items), world)
    lo = rank_ * base + min(rank_, rem)
    return list(range(lo, lo + base + (1 if rank_ < rem else 0)))


def _host_staged():
    """gloo moves only broadcast / all-reduce on device tensors; its all-gather needs host tensors (CPU rehearsals, tests)."""
    return dist.get_backend() == "gloo"


def all_gather_variable(t):
    """All-gathers tensors that differ in their first dimension; returns the concatenation in rank order."""
    if not collectives_on():
        return t
    dev = t.device
    if _host_staged():
        t = t.cpu()
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(world_size())]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    cap = max(sizes)
    pad = torch.zeros((cap,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[:t.shape[0]] = t
    out = [torch.empty_like(pad) for _ in range(world_size())]
    dist.all_gather(out, pad)
    return torch.cat([o[:s] for o, s in zip(out, sizes)], dim=0).to(dev)


def all_gather_examples(planes, pi, value):
    """Replay exchange of one iteration.  Planes travel as uint8 (they are 0/1), 4x fewer bytes than FP32."""
    if world_size() == 1:
        return planes, pi, value
    p8 = all_gather_variable(planes.to(torch.uint8))
    return p8.to(torch.float32), all_gather_variable(pi), all_gather_variable(value)


def all_gather_bytes(flat):
    """All-gathers one uint8 vector per rank (lengths differ); returns the list of every rank's vector in rank order."""
    if not collectives_on():
        return [flat]
    dev = flat.device
    if _host_staged():
        flat = flat.cpu()
    n = torch.tensor([flat.numel()], dtype=torch.int64, device=flat.device)
    sizes = [torch.zeros_like(n) for _ in range(world_size())]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    cap = (max(sizes) + 15) & ~15
    pad = torch.zeros(cap, dtype=torch.uint8, device=flat.device)
    pad[:flat.numel()] = flat
    out = torch.empty(world_size() * cap, dtype=torch.uint8, device=flat.device)
    dist.all_gather_into_tensor(out, pad) if not _host_staged() else dist.all_gather(list(out.view(world_size(), cap).unbind(0)), pad)
    return [out[r * cap:r * cap + s].to(dev) for r, s in enumerate(sizes)]


last_exchange = {}  # bytes this rank sent / received and the wall milliseconds of the latest all_gather_packed (bench.py --coach-iter)


def all_gather_packed(replay):
    """Replay exchange of one iteration: every rank's PackedReplay (replay.py: ~0.4 KB per example -- key,
    item sizes, sparse visit counts, value, episode / move) travels as ONE flat byte vector; the result is the union, sorted into the
    reference's order (episode by episode, move by move), identical on every rank."""
    import time
    from .replay import PackedReplay
    if not collectives_on():
        last_exchange.update(bytes_sent=0, bytes_received=0, ms=0.0, examples=len(replay))
        return replay
    if replay.device.type == "cuda":
        torch.cuda.synchronize(replay.device)
    t0 = time.time()
    flat = replay.to_flat()
    parts = all_gather_bytes(flat)
    out = PackedReplay.cat([PackedReplay.from_flat(p, replay.W, replay.H, replay.N) for p in parts]).sort_by_episode_move()
    if replay.device.type == "cuda":
        torch.cuda.synchronize(replay.device)
    last_exchange.update(bytes_sent=int(flat.numel()), bytes_received=int(sum(p.numel() for p in parts)), ms=(time.time() - t0) * 1e3, examples=len(out))
    return out


class FlatGradAllReduce:
    """Gradient hook for NNetWrapper.train_tensors: one flat FP32 bucket, one all-reduce(sum).  Every rank's loss is already
    divided by the FULL batch size, so the sum over ranks is the gradient of the whole batch.  Scalars passed as `extra`
    (the two loss values) ride in the same message and come back summed."""

    def __init__(self, module):
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        self.flat = None
        self.timing = None  # a list: (start event, end event) around every all-reduce (device tensors only; bench.py --coach-iter)
        self.calls = 0

    def __call__(self, module, extra=()):
        extra = tuple(extra)
        if not collectives_on():
            return extra
        dev = self.params[0].device
        total = self.numel + len(extra)
        if self.flat is None or self.flat.device != dev or self.flat.numel() != total:
            self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                self.flat[off:off + n].zero_()
            else:
                self.flat[off:off + n].copy_(p.grad.reshape(-1))
            off += n
        for k, x in enumerate(extra):
            self.flat[off + k] = x
        timed = self.timing is not None and dev.type == "cuda"
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        if timed:
            e1.record()
            self.timing.append((e0, e1))
        self.calls += 1
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                p.grad = torch.empty_like(p)
            p.grad.copy_(self.flat[off:off + n].view_as(p))
            off += n
        return tuple(self.flat[off + k].clone() for k in range(len(extra)))


def broadcast_parameters(module, src=0):
    if not collectives_on():
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src)


def attach(nnet_wrapper):
    """Makes `nnet_wrapper.train*` data-parallel: identical initial weights, gradients summed over the ranks' batch slices."""
    broadcast_parameters(nnet_wrapper.nnet)
    nnet_wrapper.grad_hook = FlatGradAllReduce(nnet_wrapper.nnet)
    return nnet_wrapper
