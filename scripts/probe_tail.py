"""GPU probe: the evaluator after the stem (PyTorch-ROCm, FP32) in NCHW vs channels_last, whole and per convolution."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from resource_packing_self_play_amd.binpacking.pytorch.BinpackingNNet import BinPackingNNet

class G:
    def getBoardSize(s): return (20, 20)
    def getActionSize(s): return 640
class A: num_items = 32; num_bins = 1

torch.backends.cudnn.benchmark = True
torch.backends.cudnn.allow_tf32 = False
torch.manual_seed(0)
net = BinPackingNNet(G(), A()).cuda().eval()

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t) / n * 1e6

for B in (4096, 10923):
    y = torch.randn(B, 16, 10, 10, device="cuda")
    with torch.no_grad():
        t_nchw = timeit(lambda: net.forward_from_stem(y))
        ycl = y.contiguous(memory_format=torch.channels_last)
        netcl = net.to(memory_format=torch.channels_last)
        t_cl = timeit(lambda: netcl.forward_from_stem(ycl))
        net.to(memory_format=torch.contiguous_format)
        print(f"B={B}: tail forward nchw {t_nchw:.0f} us  channels_last {t_cl:.0f} us", flush=True)
        shapes = [("S0 16->16 @10x10", 16, 16, 10), ("16->32 @10x10", 16, 32, 10), ("S1 32->32 @5x5", 32, 32, 5), ("S2 32->32 @3x3", 32, 32, 3)]
        for name, ci, co, s in shapes:
            w = torch.randn(co, ci, 3, 3, device="cuda"); x = torch.randn(B, ci, s, s, device="cuda")
            t1 = timeit(lambda: F.conv2d(x, w, None, padding=1))
            xcl = x.contiguous(memory_format=torch.channels_last); wcl = w.contiguous(memory_format=torch.channels_last)
            t2 = timeit(lambda: F.conv2d(xcl, wcl, None, padding=1))
            fl = 2 * B * ci * co * 9 * s * s
            print(f"   {name}: nchw {t1:.0f} us ({fl/t1/1e6:.1f} TF)  nhwc {t2:.0f} us ({fl/t2/1e6:.1f} TF)", flush=True)
        x = torch.randn(B, 288, device="cuda"); w = torch.randn(288, 288, device="cuda")
        t3 = timeit(lambda: torch.mm(x, w)); print(f"   mm [B,288]x[288,288]: {t3:.0f} us ({2*B*288*288/t3/1e6:.1f} TF)")
        x = torch.randn(B, 800, device="cuda"); w = torch.randn(800, 800, device="cuda")
        t4 = timeit(lambda: torch.mm(x, w)); print(f"   mm [B,800]x[800,800]: {t4:.0f} us ({2*B*800*800/t4/1e6:.1f} TF; vs direct 5x5 conv flops {2*B*32*32*9*25/t4/1e6:.1f} TF-equivalent)")
        x = torch.randn(B, 1600, device="cuda"); w = torch.randn(1600, 1600, device="cuda")
        t5 = timeit(lambda: torch.mm(x, w)); print(f"   mm [B,1600]x[1600,1600]: {t5:.0f} us ({2*B*1600*1600/t5/1e6:.1f} TF; direct 10x10x16 conv-equivalent {2*B*16*16*9*100/t5/1e6:.1f} TF)")
