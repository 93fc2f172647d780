"""GPU probe: rp_nn_resblock16 against the MIOpen + fused element-wise form of the same residual block, standalone."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from resource_packing_self_play_amd import _lib

torch.backends.cudnn.benchmark = True
eng = _lib.Engine(20, 20, 32, 1, 1, stream=torch.cuda.current_stream().cuda_stream)

def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t) / n * 1e6

for B in (10923, 30000, 32768):
    x = torch.randn(B, 16, 10, 10, device="cuda").contiguous(memory_format=torch.channels_last)
    w0 = torch.randn(16, 16, 3, 3, device="cuda").contiguous(memory_format=torch.channels_last) * 0.1
    w1 = torch.randn(16, 16, 3, 3, device="cuda").contiguous(memory_format=torch.channels_last) * 0.1
    b0 = torch.randn(16, device="cuda"); b1 = torch.randn(16, device="cuda")
    f0 = torch.empty(36 * 64, device="cuda"); f1 = torch.empty(36 * 64, device="cuda")
    eng.nn_pack_conv16(w0.contiguous(), f0); eng.nn_pack_conv16(w1.contiguous(), f1)
    out, out_r = torch.empty_like(x), torch.empty_like(x)
    xr = torch.relu(x)
    def miopen_form():
        c0 = F.conv2d(xr, w0, None, padding=1); eng.nn_bias_relu(c0, b0)
        c1 = F.conv2d(c0, w1, None, padding=1); eng.nn_bias_residual(c1, b1, x, out, out_r)
    def fused_form():
        eng.nn_resblock16(x, f0, b0, f1, b1, out, out_r)
    f4 = torch.cat([f0, f1, f0, f1]); b4 = torch.cat([b0, b1, b0, b1])
    def stage_form():
        eng.nn_resstage16(x, f4, b4, out, None)
    t_c = timeit(stage_form)
    print(f"B={B}: rp_nn_resstage16 (two blocks) {t_c:.0f} us ({2*2*2*B*16*16*9*100/t_c/1e6:.1f} TF)", flush=True)
    t_a = timeit(miopen_form); t_b = timeit(fused_form)
    fl = 2 * 2 * B * 16 * 16 * 9 * 100
    print(f"B={B}: MIOpen+fused-elementwise {t_a:.0f} us ({fl/t_a/1e6:.1f} TF)   rp_nn_resblock16 {t_b:.0f} us ({fl/t_b/1e6:.1f} TF)", flush=True)

for (B, H) in ((10923, 5), (10923, 3)):
    x = torch.randn(B, 32, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
    ws = [torch.randn(32, 32, 3, 3, device="cuda") * 0.05 for _ in range(4)]
    f4 = torch.empty(4 * 36 * 64 * 4, device="cuda"); b4 = torch.randn(128, device="cuda")
    for k, w in enumerate(ws):
        eng.nn_pack_conv32(w.contiguous(), f4[k * 9216:(k + 1) * 9216])
    out = torch.empty_like(x)
    t = timeit(lambda: eng.nn_resstage32(x, f4, b4, out, None))
    print(f"B={B} {H}x{H}x32: rp_nn_resstage32 (two blocks) {t:.0f} us ({4*2*B*32*32*9*H*H/t/1e6:.1f} TF)", flush=True)

for (cin, H) in ((16, 10), (32, 5)):
    B = 10923
    x = torch.randn(B, cin, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
    w = (torch.randn(32, cin, 3, 3, device="cuda") * 0.05).contiguous(); b = torch.randn(32, device="cuda")
    f = torch.empty(9 * cin * 32, device="cuda"); eng.nn_pack_conv32(w, f)
    out = torch.empty(B, 32, (H + 1) // 2, (H + 1) // 2, device="cuda").contiguous(memory_format=torch.channels_last)
    t = timeit(lambda: eng.nn_convpool32(x, f, b, out))
    print(f"B={B} {H}x{H}x{cin}->32 + pool: rp_nn_convpool32 {t:.0f} us ({2*B*cin*32*9*H*H/t/1e6:.1f} TF)", flush=True)
