"""GPU probe: the 32-channel stage kernels alone.  usage: probe_stage32.py <kernel: rs5 | rs3 | cp16 | cp32> [B] [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from resource_packing_self_play_amd import _lib
which = sys.argv[1] if len(sys.argv) > 1 else "rs5"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 30
eng = _lib.Engine(20, 20, 32, 1, 1, stream=torch.cuda.current_stream().cuda_stream)
torch.manual_seed(0)
if which in ("rs5", "rs3"):
    H = 5 if which == "rs5" else 3
    x = torch.randn(B, 32, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
    f4 = torch.empty(4 * 9216, device="cuda"); b4 = torch.randn(128, device="cuda")
    for k in range(4):
        eng.nn_pack_conv32((torch.randn(32, 32, 3, 3, device="cuda") * 0.05).contiguous(), f4[k * 9216:(k + 1) * 9216])
    out = torch.empty_like(x)
    fn = lambda: eng.nn_resstage32(x, f4, b4, out, None)
    flops = 4 * 2 * 9 * 32 * 32 * H * H * B
else:
    cin, H = (16, 10) if which == "cp16" else (32, 5)
    x = torch.randn(B, cin, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
    w = (torch.randn(32, cin, 3, 3, device="cuda") * 0.05).contiguous(); b = torch.randn(32, device="cuda")
    f = torch.empty(9 * cin * 32, device="cuda"); eng.nn_pack_conv32(w, f)
    out = torch.empty(B, 32, (H + 1) // 2, (H + 1) // 2, device="cuda").contiguous(memory_format=torch.channels_last)
    fn = lambda: eng.nn_convpool32(x, f, b, out)
    flops = 2 * 9 * cin * 32 * H * H * B
for _ in range(5): fn()
torch.cuda.synchronize(); t = time.time()
for _ in range(iters): fn()
torch.cuda.synchronize(); dt = (time.time() - t) / iters
print("%s B=%d: %.1f us per launch, %.1f TFLOP/s (%.3f of 157.3)" % (which, B, dt * 1e6, flops / dt / 1e12, flops / dt / 1e12 / 157.3))

if os.environ.get("CP_STAMP") and which in ("cp16", "cp32"):  # RP_ENGINE_LIB = a -DCP_STAMP build with the rp_debug_cp_stamp export
    import ctypes
    import numpy as np
    st = np.zeros(8, dtype=np.uint64)
    torch.cuda.synchronize()
    eng.L.rp_debug_cp_stamp(st.ctypes.data_as(ctypes.c_void_p), 1)
    fn(); torch.cuda.synchronize()
    eng.L.rp_debug_cp_stamp(st.ctypes.data_as(ctypes.c_void_p), 1)
    st = st.astype(float)
    names = ["stage x + sync", "conv", "sync + staging + sync", "pooling", "sync + re-zero"]
    print("per leaf, s_memtime ticks summed over waves / leaves:")
    for k in range(5):
        print("  %-22s %8.1f" % (names[k], st[k] / B))
    print("  total                  %8.1f" % (st[:5].sum() / B))
