"""ctypes binding of librp_engine.so (include/rp_engine.h).

There is no CPU fallback: if the HIP library is missing or no MI355X is visible,
constructing an engine raises.  Build with `python __graft_entry__.py build` or
`make -C resource_packing_self_play_amd/csrc`.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# RP_ENGINE_LIB: another build of the same library (kernel experiments with different compile-time settings); must exist
LIB_PATH = os.environ.get("RP_ENGINE_LIB") or os.path.join(HERE, "csrc", "librp_engine.so")
ABI_VERSION = 4

MOVE_EXTERNAL, MOVE_ARGMAX_FIRST, MOVE_SAMPLE = 0, 1, 2
PHASE_IDLE, PHASE_RUNNING, PHASE_WAIT_EVAL, PHASE_MOVE_READY, PHASE_EPISODE_DONE, PHASE_FAILED = range(6)
KIND_WEAK, KIND_F32, KIND_F64 = 0, 1, 2
ERR_ARG, ERR_DEVICE, ERR_CAPACITY, ERR_ASSERT, ERR_STATE = -1, -2, -3, -4, -5
COUNTER_NAMES = ("simulations", "expansions", "terminal_returns", "path_edges", "sum_valid_select", "sum_valid_leaf",
                 "transposition_links", "nodes", "moves", "episodes", "hash_probes", "key_bytes", "sum_visited_select", "visited_new",
                 "reserved0", "reserved1")


class RpConfig(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("W", C.c_int32), ("H", C.c_int32), ("N", C.c_int32), ("games", C.c_int32),
                ("sims", C.c_int32), ("cpuct", C.c_double), ("alpha", C.c_double), ("node_cap", C.c_int32),
                ("edge_cap", C.c_int32), ("move_rule", C.c_int32), ("auto_restart", C.c_int32), ("reclaim", C.c_int32),
                ("reserved0", C.c_int32), ("seed", C.c_uint64),
                ("tie_salt", C.c_uint64), ("device", C.c_int32), ("vis_cap", C.c_int32), ("stream", C.c_void_p),
                ("max_examples", C.c_int64), ("max_sparse", C.c_int64)]


class EngineError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("rp_engine error %d: %s" % (code, msg))
        self.code = code


_vp, _i32, _i64, _f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_double
_SIGS = {
    "rp_version": (C.c_int, []),
    "rp_create": (C.c_int, [C.POINTER(RpConfig), C.POINTER(_vp)]),
    "rp_destroy": (None, [_vp]),
    "rp_last_error": (C.c_char_p, [_vp]),
    "rp_device_bytes": (_i64, [_vp]),
    "rp_valid_moves": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "rp_apply_move": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rp_game_ended": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _f64, _vp, _vp]),
    "rp_begin_episodes": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp]),
    "rp_set_instance_pool": (C.c_int, [_vp, _i64, _vp, _vp, C.c_uint64]),
    "rp_begin_pool": (C.c_int, [_vp]),
    "rp_generate_items": (C.c_int, [_vp, _i64, _vp, _i32, _i32, _vp]),
    "rp_set_instance_pool_seeds": (C.c_int, [_vp, _i64, _vp, _i32, _i32, C.c_uint64]),
    "rp_set_rank_buffer": (C.c_int, [_vp, _vp, _i32]),
    "rp_set_roots": (C.c_int, [_vp, _i32, _i32, _vp, _vp]),
    "rp_set_stream": (C.c_int, [_vp, _vp]),
    "rp_set_step_cap": (C.c_int, [_vp, _i32]),
    "rp_set_compact_rows": (C.c_int, [_vp, _i32]),
    "rp_set_move_rule": (C.c_int, [_vp, _i32, _i32]),
    "rp_set_sims": (C.c_int, [_vp, _i32]),
    "rp_last_values": (C.c_int, [_vp, _i32, _i32, _vp, _vp]),
    "rp_search_step": (C.c_int, [_vp, _vp]),
    "rp_leaf_planes": (C.c_int, [_vp, _vp, _i64]),
    "rp_stem_set_weights": (C.c_int, [_vp, _vp, _vp]),
    "rp_leaf_stem": (C.c_int, [_vp, _vp, _vp, _i64, _i32]),
    "rp_nn_pack_conv16": (C.c_int, [_vp, _vp, _vp]),
    "rp_nn_value_head": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i32]),
    "rp_nn_resblock16": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32]),
    "rp_nn_resstage16": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32]),
    "rp_nn_pack_conv32": (C.c_int, [_vp, _vp, _vp, _i32]),
    "rp_nn_convpool32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32]),
    "rp_nn_resstage32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32]),
    "rp_nn_bias_relu": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i32]),
    "rp_nn_bias_residual": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32]),
    "rp_nn_bias_pool": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32]),
    "rp_leaf_states": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp]),
    "rp_commit_eval": (C.c_int, [_vp, _vp, _vp]),
    "rp_commit_eval_logits": (C.c_int, [_vp, _vp, _vp]),
    "rp_commit_eval_host": (C.c_int, [_vp, _vp, _vp, _i32]),
    "rp_root_counts": (C.c_int, [_vp, _i32, _i32, _vp]),
    "rp_game_status": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp, _vp]),
    "rp_advance_roots": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp]),
    "rp_pop_finished": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "rp_counters": (C.c_int, [_vp, _vp, _i32]),
    "rp_examples_count": (C.c_int, [_vp, _vp]),
    "rp_examples_tensors": (C.c_int, [_vp, _i64, _i64, _vp, _vp, _vp]),
    "rp_examples_meta": (C.c_int, [_vp, _i64, _i64, _vp, _vp]),
    "rp_examples_clear": (C.c_int, [_vp]),
    "rp_examples_packed_count": (C.c_int, [_vp, _vp, _vp]),
    "rp_examples_packed": (C.c_int, [_vp, _i64, _i64] + [_vp] * 9),
    "rp_expand_examples": (C.c_int, [_vp, _i64, _vp, _i64, _i64] + [_vp] * 10),
    "rp_check": (C.c_int, [_vp]),
    "rp_leaf_count_async": (C.c_int, [_vp, _vp]),
    "rp_tree_size": (C.c_int, [_vp, _i32, _vp, _vp]),
    "rp_arena_peak": (C.c_int, [_vp, _vp, _vp, _vp]),
    "rp_dump_tree": (C.c_int, [_vp, _i32] + [_vp] * 14),
    "rp_selftest_sqrt": (C.c_int, [_vp, _i64, _vp, _vp]),
    "rp_selftest_q_update": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rp_selftest_masked_prior": (C.c_int, [_vp, _i64, _vp, _vp, _vp]),
}
_lib = None


def load():
    """Loads the shared library and checks the ABI version.  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: build the HIP engine first (python __graft_entry__.py build); "
                              "there is no CPU fallback" % LIB_PATH)
        # PyTorch-ROCm wheels bundle their own HIP runtime (same SONAME as /opt/rocm's).  Import torch first so that
        # librp_engine.so binds to the runtime torch uses: one process must hold exactly one HIP runtime, or device
        # pointers and streams could not be shared between the engine and the evaluator.
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        _check_single_hip_runtime()
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        if L.rp_version() != ABI_VERSION:
            raise ImportError("librp_engine.so ABI %d != %d" % (L.rp_version(), ABI_VERSION))
        _lib = L
    return _lib


def _check_single_hip_runtime():
    try:
        with open("/proc/self/maps") as f:
            libs = {line.split()[-1] for line in f if "libamdhip64" in line}
    except OSError:
        return
    if len(libs) > 1:
        raise ImportError("two HIP runtimes are loaded (%s): import torch before anything that loads librp_engine.so"
                          % ", ".join(sorted(libs)))


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    return a.ctypes.data_as(C.c_void_p)


def _arr(a, dtype, shape=None):
    a = np.ascontiguousarray(a, dtype=dtype)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError("expected shape %s, got %s" % (tuple(shape), a.shape))
    return a


class Engine:
    """Thin object wrapper of one rp_ctx (one per process and GPU)."""

    def __init__(self, W, H, N, games, sims, cpuct=1.0, alpha=0.75, move_rule=MOVE_EXTERNAL, seed=0, tie_salt=0,
                 node_cap=0, edge_cap=0, device=0, stream=0, auto_restart=0, max_examples=0, vis_cap=0, reclaim=0, max_sparse=0):
        self.L = load()
        self.W, self.H, self.N, self.A, self.G, self.sims = int(W), int(H), int(N), int(W) * int(N), int(games), int(sims)
        self.move_rule = int(move_rule)
        cfg = RpConfig(ABI_VERSION, W, H, N, games, sims, float(cpuct), float(alpha), node_cap, edge_cap, move_rule,
                       auto_restart, 1 if reclaim else 0, 0, seed, tie_salt, device, vis_cap, stream or None, max_examples, max_sparse)
        self.KW = self.H * (2 if self.W > 32 else 1) + (self.N + 31) // 32
        if self.W > 32 and self.KW % 2:
            self.KW += 1  # 64-bit rows stay 8-byte aligned (rp_engine.h: key layout)
        h = _vp()
        rc = self.L.rp_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise EngineError(rc, self.L.rp_last_error(None).decode())
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.rp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise EngineError(rc, self.L.rp_last_error(self.h).decode())

    @property
    def device_bytes(self):
        return self.L.rp_device_bytes(self.h)

    # ---- stateless rules -----------------------------------------------------------------
    def valid_moves(self, rows, remaining, item_wh):
        rows = _arr(rows, np.uint64); B = rows.shape[0]
        rows = _arr(rows, np.uint64, (B, self.H)); remaining = _arr(remaining, np.uint8, (B, self.N))
        item_wh = _arr(item_wh, np.uint8, (B, self.N, 2))
        mask = np.empty((B, self.A), np.uint8); nv = np.empty(B, np.int32)
        self._ck(self.L.rp_valid_moves(self.h, B, _ptr(rows), _ptr(remaining), _ptr(item_wh), _ptr(mask), _ptr(nv)))
        return mask, nv

    def apply_move(self, rows, remaining, item_wh, action):
        rows = _arr(rows, np.uint64); B = rows.shape[0]
        rows = _arr(rows, np.uint64, (B, self.H)); remaining = _arr(remaining, np.uint8, (B, self.N))
        item_wh = _arr(item_wh, np.uint8, (B, self.N, 2)); action = _arr(action, np.int32, (B,))
        rows_o = np.empty_like(rows); rem_o = np.empty_like(remaining); st = np.empty(B, np.int32)
        self._ck(self.L.rp_apply_move(self.h, B, _ptr(rows), _ptr(remaining), _ptr(item_wh), _ptr(action), _ptr(rows_o), _ptr(rem_o), _ptr(st)))
        return rows_o, rem_o, st

    def game_ended(self, rows, remaining, item_wh, total_area, max_h, rewards, alpha):
        rows = _arr(rows, np.uint64); B = rows.shape[0]
        rows = _arr(rows, np.uint64, (B, self.H)); remaining = _arr(remaining, np.uint8, (B, self.N))
        item_wh = _arr(item_wh, np.uint8, (B, self.N, 2))
        total_area = _arr(total_area, np.int32, (B,)); max_h = _arr(max_h, np.int32, (B,))
        rewards = _arr(rewards, np.float64)
        ended = np.empty(B, np.int32); r = np.empty(B, np.float64)
        self._ck(self.L.rp_game_ended(self.h, B, _ptr(rows), _ptr(remaining), _ptr(item_wh), _ptr(total_area), _ptr(max_h),
                                      _ptr(rewards) if rewards.size else None, rewards.size, float(alpha), _ptr(ended), _ptr(r)))
        return ended, r

    # ---- episodes ------------------------------------------------------------------------
    def set_rank_buffer(self, rewards):
        rewards = _arr(rewards, np.float64)
        self._ck(self.L.rp_set_rank_buffer(self.h, _ptr(rewards) if rewards.size else None, rewards.size))

    def begin_episodes(self, item_wh, total_area, first=0, episode_id=None):
        item_wh = _arr(item_wh, np.uint8); count = item_wh.shape[0]
        item_wh = _arr(item_wh, np.uint8, (count, self.N, 2)); total_area = _arr(total_area, np.int32, (count,))
        ids = None if episode_id is None else _arr(episode_id, np.uint64, (count,))
        self._ck(self.L.rp_begin_episodes(self.h, first, count, _ptr(item_wh), _ptr(total_area), _ptr(ids)))

    def generate_items(self, seeds, bin_w=None, bin_h=None):
        """ItemsGenerator.items_generator for every seed, on device -> uint8 [n, N, 2] (w, h)."""
        seeds = _arr(seeds, np.uint32).reshape(-1)
        out = np.empty((seeds.shape[0], self.N, 2), np.uint8)
        self._ck(self.L.rp_generate_items(self.h, seeds.shape[0], _ptr(seeds), int(bin_w or self.W), int(bin_h or self.H), _ptr(out)))
        return out

    def set_instance_pool_seeds(self, seeds, bin_w=None, bin_h=None, first_id=0):
        seeds = _arr(seeds, np.uint32).reshape(-1)
        self._ck(self.L.rp_set_instance_pool_seeds(self.h, seeds.shape[0], _ptr(seeds), int(bin_w or self.W), int(bin_h or self.H), int(first_id)))

    def set_roots(self, rows, remaining, first=0):
        rows = _arr(rows, np.uint64); count = rows.shape[0]
        rows = _arr(rows, np.uint64, (count, self.H)); remaining = _arr(remaining, np.uint8, (count, self.N))
        self._ck(self.L.rp_set_roots(self.h, first, count, _ptr(rows), _ptr(remaining)))

    def set_compact_rows(self, enable=True):
        self._ck(self.L.rp_set_compact_rows(self.h, 1 if enable else 0))

    def set_step_cap(self, max_sims_per_step):
        self._ck(self.L.rp_set_step_cap(self.h, int(max_sims_per_step)))

    def set_stream(self, stream_handle):
        self._ck(self.L.rp_set_stream(self.h, C.c_void_p(stream_handle or None)))

    def set_move_rule(self, move_rule, onehot_examples=False):
        self._ck(self.L.rp_set_move_rule(self.h, int(move_rule), 1 if onehot_examples else 0))
        self.move_rule = int(move_rule)

    def set_sims(self, sims):
        self._ck(self.L.rp_set_sims(self.h, int(sims)))
        self.sims = int(sims)

    def last_values(self, first=0, count=None):
        count = self.G - first if count is None else count
        v = np.empty(count, np.float64); k = np.empty(count, np.int32)
        self._ck(self.L.rp_last_values(self.h, first, count, _ptr(v), _ptr(k)))
        return v, k

    # ---- search --------------------------------------------------------------------------
    def search_step(self, sync=True):
        if not sync:
            self._ck(self.L.rp_search_step(self.h, None))
            return None
        n = _i32(0)
        self._ck(self.L.rp_search_step(self.h, C.byref(n)))
        return n.value

    def leaf_planes(self, dev_ptr, capacity_rows):
        self._ck(self.L.rp_leaf_planes(self.h, C.c_void_p(dev_ptr), capacity_rows))

    def stem_set_weights(self, conv_w_dev_ptr, bias_dev_ptr):
        self._ck(self.L.rp_stem_set_weights(self.h, C.c_void_p(conv_w_dev_ptr), C.c_void_p(bias_dev_ptr)))

    def leaf_stem(self, dev_ptr, capacity_rows, relu_dev_ptr=None, channels_last=False):
        self._ck(self.L.rp_leaf_stem(self.h, C.c_void_p(dev_ptr), C.c_void_p(relu_dev_ptr) if relu_dev_ptr else None, capacity_rows,
                                     1 if channels_last else 0))

    # fused element-wise evaluator pieces on torch tensors (float32 on this context's device, NCHW-contiguous or channels-last)
    @staticmethod
    def _bchw(x):
        """(rows, channels, inner) such that element e has channel (e // inner) % channels in x's memory order."""
        if x.dim() == 4 and not x.is_contiguous():
            import torch
            if not x.is_contiguous(memory_format=torch.channels_last):
                raise ValueError("tensor must be contiguous or channels-last")
            return x.shape[0] * x.shape[2] * x.shape[3], x.shape[1], 1
        if not x.is_contiguous():
            raise ValueError("tensor must be contiguous")
        return x.shape[0], x.shape[1], x[0, 0].numel()

    kernel_events = None  # measurement aid: a list -> (name, start event, end event) of every fused evaluator kernel launched

    def _timed(self, name, call):
        sink = self.kernel_events
        if sink is None:
            return call()
        import torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()  # torch's current stream: the one this engine was created on, in BatchedSelfPlay's groups
        call()
        e1.record()
        sink.append((name, e0, e1))

    def nn_value_head(self, z, weight, bias, out):
        """out[b] = tanh(z[b] . weight + bias): z contiguous float32 [B, K], weight [1, K] or [K], bias [1], out [B] or [B, 1]."""
        B, K = z.shape
        if not (z.is_contiguous() and weight.is_contiguous() and out.is_contiguous()):
            raise ValueError("nn_value_head needs contiguous tensors")
        self._ck(self.L.rp_nn_value_head(self.h, C.c_void_p(z.data_ptr()), C.c_void_p(weight.data_ptr()), C.c_void_p(bias.data_ptr()),
                                         C.c_void_p(out.data_ptr()), B, K))

    def nn_pack_conv16(self, weight, frag):
        """weight: contiguous float32 [16, 16, 3, 3]; frag: float32 [36 * 64] buffer to fill (MFMA B-fragment order)."""
        self._ck(self.L.rp_nn_pack_conv16(self.h, C.c_void_p(weight.data_ptr()), C.c_void_p(frag.data_ptr())))

    def nn_resblock16(self, x, frag0, bias0, frag1, bias1, out, out_relu=None):
        """Fused 16-channel residual block on channels-last x [B, 16, H, W]."""
        B, Cc, H, W = x.shape
        if Cc != 16 or self._bchw(x)[2] != 1:
            raise ValueError("nn_resblock16 needs a channels-last [B, 16, H, W] tensor")
        self._ck(self.L.rp_nn_resblock16(self.h, C.c_void_p(x.data_ptr()), C.c_void_p(frag0.data_ptr()), C.c_void_p(bias0.data_ptr()),
                                         C.c_void_p(frag1.data_ptr()), C.c_void_p(bias1.data_ptr()), C.c_void_p(out.data_ptr()),
                                         C.c_void_p(out_relu.data_ptr()) if out_relu is not None else None, B, H, W))

    def nn_resstage16(self, x, frag4, bias4, out, out_relu=None):
        """Both residual blocks of a 16-channel stage on channels-last x [B, 16, H, W] (H * W <= 640) in one launch."""
        B, Cc, H, W = x.shape
        if Cc != 16 or self._bchw(x)[2] != 1 or H * W > 640:
            raise ValueError("nn_resstage16 needs a channels-last [B, 16, H, W] tensor with H * W <= 640")
        self._timed("k_resstage16 %dx%d" % (H, W), lambda: self._ck(self.L.rp_nn_resstage16(
            self.h, C.c_void_p(x.data_ptr()), C.c_void_p(frag4.data_ptr()), C.c_void_p(bias4.data_ptr()), C.c_void_p(out.data_ptr()),
            C.c_void_p(out_relu.data_ptr()) if out_relu is not None else None, B, H, W)))

    def nn_pack_conv32(self, weight, frag):
        """weight: contiguous float32 [32, Cin, 3, 3], Cin 16 or 32; frag: float32 [9 * Cin * 32] buffer to fill (streaming
        B-fragment order)."""
        self._ck(self.L.rp_nn_pack_conv32(self.h, C.c_void_p(weight.data_ptr()), C.c_void_p(frag.data_ptr()), int(weight.shape[1])))

    def nn_convpool32(self, x, frag, bias, out):
        """conv3x3(Cin -> 32) + bias + max_pool2d(3, 2, 1) on channels-last x [B, Cin, H, W] -> out [B, 32, (H+1)//2, (W+1)//2]."""
        B, Cc, H, W = x.shape
        if self._bchw(x)[2] != 1 or self._bchw(out)[2] != 1:
            raise ValueError("nn_convpool32 needs channels-last tensors")
        self._timed("k_convpool32 %d->32 %dx%d" % (Cc, H, W), lambda: self._ck(self.L.rp_nn_convpool32(
            self.h, C.c_void_p(x.data_ptr()), C.c_void_p(frag.data_ptr()), C.c_void_p(bias.data_ptr()), C.c_void_p(out.data_ptr()), B, Cc, H, W)))

    def nn_resstage32(self, x, frag4, bias4, out, out_relu=None):
        """Both residual blocks of a 32-channel stage on channels-last x [B, 32, H, W] (H * W <= 512) in one launch."""
        B, Cc, H, W = x.shape
        if Cc != 32 or self._bchw(x)[2] != 1 or H * W > 512:
            raise ValueError("nn_resstage32 needs a channels-last [B, 32, H, W] tensor with H * W <= 512")
        self._timed("k_resstage32 %dx%d" % (H, W), lambda: self._ck(self.L.rp_nn_resstage32(
            self.h, C.c_void_p(x.data_ptr()), C.c_void_p(frag4.data_ptr()), C.c_void_p(bias4.data_ptr()), C.c_void_p(out.data_ptr()),
            C.c_void_p(out_relu.data_ptr()) if out_relu is not None else None, B, H, W)))

    def nn_bias_relu(self, x, bias):
        B, Cc, inner = self._bchw(x)
        self._ck(self.L.rp_nn_bias_relu(self.h, C.c_void_p(x.data_ptr()), C.c_void_p(bias.data_ptr()), B, Cc, inner))
        return x

    def nn_bias_residual(self, x, bias, res, out, out_relu=None):
        B, Cc, inner = self._bchw(x)
        self._ck(self.L.rp_nn_bias_residual(self.h, C.c_void_p(x.data_ptr()), C.c_void_p(bias.data_ptr()), C.c_void_p(res.data_ptr()),
                                            C.c_void_p(out.data_ptr()), C.c_void_p(out_relu.data_ptr()) if out_relu is not None else None,
                                            B, Cc, inner))

    def nn_bias_pool(self, x, bias, out, out_relu=None):
        B, Cc, H, W = x.shape
        cl = self._bchw(x)[2] == 1 and H * W > 1
        self._ck(self.L.rp_nn_bias_pool(self.h, C.c_void_p(x.data_ptr()), C.c_void_p(bias.data_ptr()), C.c_void_p(out.data_ptr()),
                                        C.c_void_p(out_relu.data_ptr()) if out_relu is not None else None, B, Cc, H, W, 1 if cl else 0))

    def leaf_count_async(self, pinned_int32, index=0):
        """Enqueues a copy of the device-side count of waiting leaves into element `index` of a PINNED int32 torch tensor."""
        self._ck(self.L.rp_leaf_count_async(self.h, C.c_void_p(pinned_int32.data_ptr() + 4 * int(index))))

    # ---- replay buffer, packed (torch tensors on this context's device) -----------------------------
    def examples_packed_count(self):
        n, s = _i64(0), _i64(0)
        self._ck(self.L.rp_examples_packed_count(self.h, C.byref(n), C.byref(s)))
        return n.value, s.value

    def examples_packed(self, device):
        """Everything recorded so far as a dict of device tensors in the packed layout of rp_examples_packed (+ episode, move)."""
        import torch
        E, S = self.examples_packed_count()
        t = dict(key=torch.empty((E, self.KW), dtype=torch.int32, device=device), wh=torch.empty((E, 2 * self.N), dtype=torch.uint8, device=device),
                 value=torch.empty(E, dtype=torch.int32, device=device), sp_off=torch.empty(E, dtype=torch.int32, device=device),
                 sp_n=torch.empty(E, dtype=torch.int32, device=device), sp_act=torch.empty(S, dtype=torch.int16, device=device),
                 sp_cnt=torch.empty(S, dtype=torch.int32, device=device), episode=torch.empty(E, dtype=torch.int64, device=device),
                 move=torch.empty(E, dtype=torch.int32, device=device))
        torch.cuda.synchronize(device)  # the tensors come from the caller's stream, the copies run on the context's
        self._ck(self.L.rp_examples_packed(self.h, E, S, *[C.c_void_p(t[k].data_ptr()) for k in
                                                           ("key", "wh", "value", "sp_off", "sp_n", "sp_act", "sp_cnt", "episode", "move")]))
        # the copies run on the CONTEXT's stream; the caller's next torch operation on these tensors runs on torch's current stream, which
        # does not wait for it (torch's streams are non-blocking): hand the tensors over only once the copies have finished
        self.check()
        return t

    def expand_examples(self, index, key, wh, value, sp_off, sp_n, sp_act, sp_cnt, planes, pi, value_out):
        """rp_expand_examples on torch tensors (all on this context's device; index int64 or None)."""
        n = planes.shape[0]
        ptr = lambda x: C.c_void_p(x.data_ptr()) if x is not None else None
        self._ck(self.L.rp_expand_examples(self.h, n, ptr(index), int(key.shape[0]), int(sp_act.shape[0]), ptr(key), ptr(wh), ptr(value), ptr(sp_off),
                                           ptr(sp_n), ptr(sp_act), ptr(sp_cnt), ptr(planes), ptr(pi), ptr(value_out)))

    def check(self):
        """Synchronises the context's stream and raises if a kernel recorded a device error since the last check."""
        self._ck(self.L.rp_check(self.h))

    def leaf_states(self, max_rows=None):
        max_rows = self.G if max_rows is None else max_rows
        rows = np.empty((max_rows, self.H), np.uint64); rem = np.empty((max_rows, self.N), np.uint8)
        slot = np.empty(max_rows, np.int32); n = _i32(0)
        self._ck(self.L.rp_leaf_states(self.h, max_rows, _ptr(rows), _ptr(rem), _ptr(slot), C.byref(n)))
        return rows[:n.value], rem[:n.value], slot[:n.value]

    def commit_eval(self, pi_dev_ptr, v_dev_ptr):
        self._ck(self.L.rp_commit_eval(self.h, C.c_void_p(pi_dev_ptr), C.c_void_p(v_dev_ptr)))

    LOGITS_MAX_ACTIONS = 1536  # rp_commit_eval_logits keeps a row in LDS

    def commit_eval_logits(self, logits_dev_ptr, v_dev_ptr):
        self._ck(self.L.rp_commit_eval_logits(self.h, C.c_void_p(logits_dev_ptr), C.c_void_p(v_dev_ptr)))

    def commit_eval_host(self, pi, v):
        pi = _arr(pi, np.float32); n = pi.shape[0]
        pi = _arr(pi, np.float32, (n, self.A)); v = _arr(v, np.float32).reshape(-1)
        if v.shape[0] != n:
            raise ValueError("v must have one entry per pi row")
        self._ck(self.L.rp_commit_eval_host(self.h, _ptr(pi), _ptr(v), n))

    def run_host(self, evaluate, max_steps=10 ** 9):
        """Runs search steps with a HOST evaluator `evaluate(rows[n,H], rem[n,N]) -> (pi[n,A], v[n])` until no
        slot waits for an evaluation (all slots MOVE_READY / done).  Test and plumbing path."""
        steps = 0
        while steps < max_steps:
            n = self.search_step()
            if n == 0:
                busy = (PHASE_RUNNING,) if self.move_rule == MOVE_EXTERNAL else (PHASE_RUNNING, PHASE_MOVE_READY)
                if not np.isin(self.status()[0], busy).any():
                    return steps
                continue
            rows, rem, _ = self.leaf_states(n)
            pi, v = evaluate(rows, rem)
            self.commit_eval_host(pi, v)
            steps += 1
        return steps

    # ---- results -------------------------------------------------------------------------
    def root_counts(self, first=0, count=None):
        count = self.G - first if count is None else count
        out = np.empty((count, self.A), np.uint32)
        self._ck(self.L.rp_root_counts(self.h, first, count, _ptr(out)))
        return out

    def status(self, first=0, count=None):
        count = self.G - first if count is None else count
        ph = np.empty(count, np.int32); sd = np.empty(count, np.int32); mv = np.empty(count, np.int32); ep = np.empty(count, np.uint64)
        self._ck(self.L.rp_game_status(self.h, first, count, _ptr(ph), _ptr(sd), _ptr(mv), _ptr(ep)))
        return ph, sd, mv, ep

    def advance_roots(self, action, first=0):
        action = _arr(action, np.int32).reshape(-1); count = action.shape[0]
        ended = np.empty(count, np.int32); score = np.empty(count, np.float64)
        self._ck(self.L.rp_advance_roots(self.h, first, count, _ptr(action), _ptr(ended), _ptr(score)))
        return ended, score

    def pop_finished(self, max_n=1 << 20):
        ids = np.empty(max_n, np.uint64); oc = np.empty(max_n, np.int32); sc = np.empty(max_n, np.float64); mv = np.empty(max_n, np.int32)
        n = _i64(0)
        self._ck(self.L.rp_pop_finished(self.h, max_n, _ptr(ids), _ptr(oc), _ptr(sc), _ptr(mv), C.byref(n)))
        k = n.value
        return ids[:k], oc[:k], sc[:k], mv[:k]

    def counters(self, reset=False):
        out = np.zeros(16, np.int64)
        self._ck(self.L.rp_counters(self.h, _ptr(out), 1 if reset else 0))
        return dict(zip(COUNTER_NAMES, out.tolist()))

    def arena_peak(self):
        """-> dict: peak chunks in use over all slots and the bytes that is, for the legal-move and the visited arenas."""
        a, b = _i32(0), _i32(0)
        ce = (_i32 * 2)()
        self._ck(self.L.rp_arena_peak(self.h, C.byref(a), C.byref(b), ce))
        return {"prior_chunks": a.value, "visited_chunks": b.value, "prior_bytes": a.value * ce[0] * 6, "visited_bytes": b.value * ce[1] * 32}

    def tree_size(self, slot):
        a, b = _i32(0), _i32(0)
        self._ck(self.L.rp_tree_size(self.h, slot, C.byref(a), C.byref(b)))
        return a.value, b.value

    def dump_tree(self, slot):
        nn, ne = self.tree_size(slot)
        d = dict(node_rows=np.zeros((nn, self.H), np.uint64), node_rem=np.zeros((nn, self.N), np.uint8), node_term=np.zeros(nn, np.int8),
                 node_term_kind=np.zeros(nn, np.uint8), node_expanded=np.zeros(nn, np.uint8), node_ns=np.zeros(nn, np.uint32),
                 node_edge_off=np.zeros(nn, np.uint32), node_n_valid=np.zeros(nn, np.uint32), edge_action=np.zeros(ne, np.uint16),
                 edge_p=np.zeros(ne, np.float64), edge_q=np.zeros(ne, np.float64), edge_nsa=np.zeros(ne, np.uint32),
                 edge_q_kind=np.zeros(ne, np.uint8), edge_child=np.zeros(ne, np.uint32))
        self._ck(self.L.rp_dump_tree(self.h, slot, *[_ptr(d[k]) for k in (
            "node_rows", "node_rem", "node_term", "node_term_kind", "node_expanded", "node_ns", "node_edge_off", "node_n_valid",
            "edge_action", "edge_p", "edge_q", "edge_nsa", "edge_q_kind", "edge_child")]))
        return d

    # ---- self tests ----------------------------------------------------------------------
    def selftest_sqrt(self, n):
        a = np.empty(n, np.float64); b = np.empty(n, np.float64)
        self._ck(self.L.rp_selftest_sqrt(self.h, n, _ptr(a), _ptr(b)))
        return a, b

    def selftest_q_update(self, q, q_kind, nsa, v, v_kind):
        q = _arr(q, np.float64); n = q.shape[0]
        q_kind = _arr(q_kind, np.uint8, (n,)); nsa = _arr(nsa, np.uint32, (n,)); v = _arr(v, np.float64, (n,)); v_kind = _arr(v_kind, np.uint8, (n,))
        qo = np.empty(n, np.float64); ko = np.empty(n, np.uint8)
        self._ck(self.L.rp_selftest_q_update(self.h, n, _ptr(q), _ptr(q_kind), _ptr(nsa), _ptr(v), _ptr(v_kind), _ptr(qo), _ptr(ko)))
        return qo, ko

    def selftest_masked_prior(self, pi, valid):
        pi = _arr(pi, np.float32); B = pi.shape[0]
        pi = _arr(pi, np.float32, (B, self.A)); valid = _arr(valid, np.uint8, (B, self.A))
        out = np.empty((B, self.A), np.float64)
        self._ck(self.L.rp_selftest_masked_prior(self.h, B, _ptr(pi), _ptr(valid), _ptr(out)))
        return out
