"""Diagnostic: where does NNetWrapper.train on the GPU leave the reference's CPU weights (tests/golden/train_c2.npz)?"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from resource_packing_self_play_amd.binpacking.BinPackingGame import BinPackingGame
from resource_packing_self_play_amd.binpacking.pytorch.NNet import NNetWrapper
from resource_packing_self_play_amd.utils import dotdict
d = np.load(os.path.join(ROOT, "tests", "golden", "train_c2.npz"))
W, H, N = int(d["W"]), int(d["H"]), int(d["N"])
def mk(cuda):
    args = dotdict(cuda=cuda, num_items=N, num_bins=1, epochs=int(d["epochs"]), batch_size=int(d["batch_size"]))
    net = NNetWrapper(BinPackingGame(W, H, N, 1), args)
    net.nnet.load_state_dict({k[3:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("i__")})
    return net
examples = [(d["planes"][k].astype(np.int64), [float(x) for x in d["pi"][k]], int(d["v"][k])) for k in range(len(d["v"]))]
# first-step gradients on the same batch
np.random.seed(int(d["np_seed"])); ids = np.random.randint(len(examples), size=int(d["batch_size"]))
grads = {}
for cuda in (False, True):
    net = mk(cuda); dev = net.device
    x = torch.as_tensor(d["planes"][ids].astype(np.float32), device=dev); tp = torch.as_tensor(d["pi"][ids].astype(np.float32), device=dev)
    tv = torch.as_tensor(d["v"][ids].astype(np.float32), device=dev)
    net.nnet.train(); op, ov = net.nnet(x)
    (net.loss_pi(tp, op) + net.loss_v(tv, ov)).backward()
    grads[cuda] = {k: p.grad.detach().cpu().numpy() for k, p in net.nnet.named_parameters()}
for k in grads[False]:
    a, b = grads[False][k], grads[True][k]
    print("grad %-40s max|g| %.2e  max|dg| %.2e  zeros cpu %d gpu %d  tiny(<1e-7) cpu %d gpu %d of %d" % (k, np.abs(a).max(), np.abs(a - b).max(), (a == 0).sum(), (b == 0).sum(),
          (np.abs(a) < 1e-7).sum(), (np.abs(b) < 1e-7).sum(), a.size))
for cuda in (False, True):
    net = mk(cuda)
    np.random.seed(int(d["np_seed"]))
    net.train(examples)
    tot = 0; bad = 0
    for k, t in net.nnet.state_dict().items():
        diff = np.abs(t.cpu().numpy() - d["f__" + k]); ref_move = np.abs(d["f__" + k] - d["i__" + k])
        nb = int((diff > 1e-4).sum()); tot += diff.size; bad += nb
        if nb:
            print("%s %-40s max %.2e  over 1e-4: %d of %d   (reference moved those by %.2e .. %.2e)" % ("gpu" if cuda else "cpu", k, diff.max(), nb, diff.size, ref_move[diff > 1e-4].min(), ref_move[diff > 1e-4].max()))
    print("gpu" if cuda else "cpu", "elements over 1e-4: %d of %d" % (bad, tot))
