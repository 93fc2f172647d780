"""Packed replay set: what `CoachBPP.learn` keeps in `trainExamplesHistory` (xw_mcts/CoachBPP.py:152-157), what the ranks
all-gather once per iteration, and what `NNetWrapper.train` samples its minibatches from (xw_mcts/binpacking/pytorch/NNet.py:39-46).

The reference stores every example as an `(N+1, H, W)` int64 state plus a length-A list of floats (105.6 KB + 5 KB at 20x20 / 32
items); dense FP32 training tensors are 55 KB.  An example here is what the engine recorded (include/rp_engine.h,
rp_examples_packed):

    key      int32 [E][KW]   rows + remaining-item words (84 B at 20x20 / 32)
    wh       uint8 [E][2N]   the episode's item sizes (64 B)
    value    int32 [E]       the episode's ranked outcome
    sp_off   int64 [E]       first entry / number of entries of the example's (action, visit count) pairs in the pool:
    sp_n     int32 [E]       one pair per VISITED root edge (20-50 of 640 actions at 400 sims)
    sp_act   int16 [S]       pool (actions < 8192, so the engine's uint16 reads as int16 unchanged)
    sp_cnt   int32 [S]

about 0.4 KB per example: 1 M examples (32 768 episodes x 31 moves) stay under 0.5 GB per rank.  Planes and the dense pi are
produced per MINIBATCH by the engine's expand kernel (rp_expand_examples), bit-identical to rp_examples_tensors.
Permuting, trimming and concatenating sets moves only the per-example arrays; the pool is shared until `compact()`.
"""
import torch

from . import _lib

_FIELDS = (("key", torch.int32), ("wh", torch.uint8), ("value", torch.int32), ("sp_off", torch.int64), ("sp_n", torch.int32),
           ("sp_act", torch.int16), ("sp_cnt", torch.int32))
_expanders = {}


def expander(W, H, N, device):
    """A one-slot engine context on torch's CURRENT stream of `device`, used only for its geometry by the stateless expand kernel."""
    stream = torch.cuda.current_stream(device).cuda_stream
    k = (int(W), int(H), int(N), device.index or 0, stream)
    if k not in _expanders:
        _expanders[k] = _lib.Engine(W, H, N, 1, 1, device=device.index or 0, stream=stream, node_cap=4, edge_cap=max(4096, W * N), vis_cap=1024)
    return _expanders[k]


class PackedReplay:
    def __init__(self, W, H, N, key, wh, value, sp_off, sp_n, sp_act, sp_cnt, episode=None, move=None):
        self.W, self.H, self.N, self.A = int(W), int(H), int(N), int(W) * int(N)
        self.key, self.wh, self.value, self.sp_off, self.sp_n = key, wh, value, sp_off.to(torch.int64), sp_n
        self.sp_act, self.sp_cnt = sp_act, sp_cnt
        self.episode, self.move = episode, move

    # ---- construction --------------------------------------------------------------------------------------------------
    @classmethod
    def empty(cls, W, H, N, KW, device):
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=device)
        return cls(W, H, N, z((0, KW), torch.int32), z((0, 2 * N), torch.uint8), z(0, torch.int32), z(0, torch.int64), z(0, torch.int32),
                   z(0, torch.int16), z(0, torch.int32), z(0, torch.int64), z(0, torch.int32))

    @classmethod
    def from_engine(cls, eng, device):
        t = eng.examples_packed(device)
        return cls(eng.W, eng.H, eng.N, t["key"], t["wh"], t["value"], t["sp_off"], t["sp_n"], t["sp_act"], t["sp_cnt"], t["episode"], t["move"])

    @staticmethod
    def cat(parts):
        """Concatenation in the given order; pool offsets of later parts are shifted."""
        parts = list(parts)
        first = parts[0]
        if len(parts) == 1:
            return first
        base, offs = 0, []
        for p in parts:
            offs.append(p.sp_off + base)
            base += p.sp_act.shape[0]
        meta = all(p.episode is not None for p in parts)
        return PackedReplay(first.W, first.H, first.N, torch.cat([p.key for p in parts]), torch.cat([p.wh for p in parts]),
                            torch.cat([p.value for p in parts]), torch.cat(offs), torch.cat([p.sp_n for p in parts]),
                            torch.cat([p.sp_act for p in parts]), torch.cat([p.sp_cnt for p in parts]),
                            torch.cat([p.episode for p in parts]) if meta else None, torch.cat([p.move for p in parts]) if meta else None)

    # ---- views ----------------------------------------------------------------------------------------------------------
    def __len__(self):
        return int(self.key.shape[0])

    @property
    def device(self):
        return self.key.device

    @property
    def nbytes(self):
        return sum(int(getattr(self, k).numel()) * getattr(self, k).element_size() for k, _ in _FIELDS)

    def select(self, index):
        """The examples `index` (int64 tensor) in that order; the pool is shared, not copied."""
        g = lambda t: t.index_select(0, index)
        return PackedReplay(self.W, self.H, self.N, g(self.key), g(self.wh), g(self.value), g(self.sp_off), g(self.sp_n), self.sp_act, self.sp_cnt,
                            g(self.episode) if self.episode is not None else None, g(self.move) if self.move is not None else None)

    def tail(self, keep):
        """The last `keep` examples: deque(maxlen=maxlenOfQueue) of CoachBPP.py:122."""
        keep = int(keep)
        if keep >= len(self):
            return self
        lo = len(self) - keep
        out = self.select(torch.arange(lo, len(self), device=self.device))
        return out.compact()

    def sort_by_episode_move(self):
        """The reference's order -- episode by episode, move by move (CoachBPP.py:80,133); the engine's buffer fills in completion
        order across slots (and ranks)."""
        k = self.episode * (self.N + 1) + self.move.to(torch.int64)
        return self.select(torch.argsort(k, stable=True))

    def compact(self):
        """Drops pool entries no example refers to (after tail / select)."""
        n = self.sp_n.to(torch.int64)
        total = int(n.sum().item()) if len(self) else 0
        if total == self.sp_act.shape[0]:
            return self
        new_off = torch.cumsum(n, 0) - n
        src = torch.repeat_interleave(self.sp_off - new_off, n) + torch.arange(total, device=self.device)
        return PackedReplay(self.W, self.H, self.N, self.key, self.wh, self.value, new_off, self.sp_n, self.sp_act.index_select(0, src),
                            self.sp_cnt.index_select(0, src), self.episode, self.move)

    # ---- expansion to training tensors ----------------------------------------------------------------------------------
    def expand(self, index=None):
        """(planes [n, N+1, H, W], pi [n, A], value [n]) float32 of the examples `index` (int64 device tensor; None = all), through
        the engine's kernel on torch's current stream: planes as getBinItem (BinPackingGame.py:118-120), pi = counts / sum in float64
        rounded to float32 (MCTS_bpp.py:51-54, NNet.py:46), value = the ranked outcome."""
        n = len(self) if index is None else int(index.shape[0])
        dev = self.device
        planes = torch.empty((n, self.N + 1, self.H, self.W), dtype=torch.float32, device=dev)
        pi = torch.empty((n, self.A), dtype=torch.float32, device=dev)
        value = torch.empty((n,), dtype=torch.float32, device=dev)
        if n:
            idx = None if index is None else index.to(torch.int64).contiguous()
            expander(self.W, self.H, self.N, dev).expand_examples(idx, self.key.contiguous(), self.wh.contiguous(), self.value.contiguous(),
                                                                  self.sp_off.contiguous(), self.sp_n.contiguous(), self.sp_act, self.sp_cnt, planes, pi, value)
        return planes, pi, value

    def dense(self):
        return self.expand(None)

    def check(self):
        """Synchronises and raises if the expand kernel met an index or a pool entry outside this set's arrays."""
        if self.device.type == "cuda":
            expander(self.W, self.H, self.N, self.device).check()

    # ---- one flat byte buffer (the all-gather payload, the on-disk form) ------------------------------------------------------
    def to_flat(self):
        """uint8 1-D tensor: int64 header [E, S, KW, has_meta], then every array padded to 8 bytes."""
        c = self.compact()
        meta = c.episode is not None
        head = torch.tensor([len(c), c.sp_act.shape[0], c.key.shape[1], 1 if meta else 0], dtype=torch.int64, device=c.device)
        arrays = [getattr(c, k).to(dt).contiguous() for k, dt in _FIELDS] + ([c.episode.to(torch.int64), c.move.to(torch.int32)] if meta else [])
        chunks = [head.view(torch.uint8)]
        for a in arrays:
            b = a.reshape(-1).view(torch.uint8)
            pad = (-b.numel()) % 8
            chunks.append(b)
            if pad:
                chunks.append(torch.zeros(pad, dtype=torch.uint8, device=c.device))
        return torch.cat(chunks)

    @classmethod
    def from_flat(cls, flat, W, H, N):
        head = flat[:32].view(torch.int64).cpu().tolist()
        E, S, KW, meta = (int(x) for x in head)
        shapes = {"key": (E, KW), "wh": (E, 2 * N), "value": (E,), "sp_off": (E,), "sp_n": (E,), "sp_act": (S,), "sp_cnt": (S,)}
        fields = list(_FIELDS) + ([("episode", torch.int64), ("move", torch.int32)] if meta else [])
        shapes.update(episode=(E,), move=(E,))
        pos, out = 32, {}
        for k, dt in fields:
            count = 1
            for d in shapes[k]:
                count *= d
            nb = count * torch.empty((), dtype=dt).element_size()
            out[k] = flat[pos:pos + nb].view(dt).reshape(shapes[k]) if nb else torch.zeros(shapes[k], dtype=dt, device=flat.device)
            pos += nb + ((-nb) % 8)
        return cls(W, H, N, out["key"], out["wh"], out["value"], out["sp_off"], out["sp_n"], out["sp_act"], out["sp_cnt"], out.get("episode"), out.get("move"))
