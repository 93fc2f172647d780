"""GPU probe: rp_nn_resstage16 alone (10x10x16, B leaves), for rocprofv3 --pmc / --kernel-trace runs.  usage: probe_stage16.py [B] [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from resource_packing_self_play_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
eng = _lib.Engine(20, 20, 32, 1, 1, stream=torch.cuda.current_stream().cuda_stream)
torch.manual_seed(0)
x = torch.randn(B, 16, 10, 10, device="cuda").contiguous(memory_format=torch.channels_last)
f4 = torch.empty(4 * 36 * 64, device="cuda"); b4 = torch.randn(64, device="cuda")
for k in range(4):
    eng.nn_pack_conv16((torch.randn(16, 16, 3, 3, device="cuda") * 0.1).contiguous(), f4[k * 2304:(k + 1) * 2304])
out = torch.empty_like(x)
for _ in range(5):
    eng.nn_resstage16(x, f4, b4, out, None)
torch.cuda.synchronize(); t = time.time()
for _ in range(iters):
    eng.nn_resstage16(x, f4, b4, out, None)
torch.cuda.synchronize(); dt = (time.time() - t) / iters
print("B=%d: %.1f us per launch, %.1f TFLOP/s (%.3f of 157.3)" % (B, dt * 1e6, 4 * 2 * 9 * 16 * 16 * 100 * B / dt / 1e12, 4 * 2 * 9 * 16 * 16 * 100 * B / dt / 1e12 / 157.3))

if os.environ.get("RS_STAMP"):
    stamp = torch.zeros_like(out); torch.cuda.synchronize()
    eng.nn_resstage16(x, f4, b4, out, stamp)
    torch.cuda.synchronize()
    st = stamp.permute(0, 2, 3, 1).reshape(-1)[:20].contiguous().view(torch.int64).cpu().numpy().astype(float)
    names = ["stage x", "sync", "conv0", "epi0+sync", "conv1", "epi1+sync", "conv2", "epi2+sync", "conv3", "epi3"]
    order = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9]
    print("per leaf, shader-clock cycles (sum over waves / leaves):")
    for k in order:
        print("  %-10s %8.0f" % (names[k], st[k] / B))
    print("  total      %8.0f" % (st.sum() / B))
