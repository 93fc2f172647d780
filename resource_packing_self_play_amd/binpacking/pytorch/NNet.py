"""`NNetWrapper` with the reference's interface (xw_mcts/binpacking/pytorch/NNet.py:17-111): `predict`, `train`,
`save_checkpoint`, `load_checkpoint`, attribute `.nnet`.  Additions for the batched engine: `predict_batch`
(device tensors in, device tensors out, feeds rp_commit_eval) and `train_tensors` (replay tensors produced on
device by rp_examples_tensors).  FP32 throughout: the parity bar against the CPU reference is 1e-5.
"""
import os

import numpy as np
import torch
from torch import optim

from ...NeuralNet import NeuralNet
from ...utils import AverageMeter
from .BinpackingNNet import BinPackingNNet


class NNetWrapper(NeuralNet):
    def __init__(self, game, args):
        self.args = args
        self.nnet = BinPackingNNet(game, args)
        self.board_h, self.board_w = game.getBoardSize()
        self.action_size = game.getActionSize()
        self.in_planes = int(args.num_items) + int(args.num_bins)
        self.device = torch.device("cuda", torch.cuda.current_device()) if getattr(args, "cuda", False) else torch.device("cpu")
        if self.device.type == "cuda":
            torch.backends.cudnn.allow_tf32 = False
            torch.backends.cudnn.benchmark = True  # let MIOpen pick its fastest FP32 convolution per shape
            torch.backends.cuda.matmul.allow_tf32 = False
            self.nnet.to(self.device)
        self.grad_hook = None  # set by distributed.attach(): all-reduces gradients before optimizer.step()
        self.log = getattr(args, "verbose", False)

    # ---- inference ---------------------------------------------------------------------------
    def predict(self, board):
        """board: numpy state (N+1, H, W) -> (pi, v) numpy float32, shapes (A,) and (1,)  (reference :69-85)"""
        x = torch.as_tensor(np.asarray(board).astype(np.float64), dtype=torch.float32, device=self.device)
        x = x.view(-1, self.in_planes, self.board_h, self.board_w)
        self.nnet.eval()
        with torch.no_grad():
            log_pi, v = self.nnet(x)
        return torch.exp(log_pi).cpu().numpy()[0], v.cpu().numpy()[0]

    def predict_batch(self, planes):
        """planes: float32 tensor [B, N+1, H, W] on self.device -> (pi [B, A], v [B]) contiguous float32 on device."""
        self.nnet.eval()
        with torch.no_grad():
            log_pi, v = self.nnet(planes)
            return torch.exp(log_pi).contiguous(), v.reshape(-1).contiguous()

    def stem_params(self):
        """(weight [16, N+1, 3, 3], bias [16]) of the first convolution, contiguous FP32 on the device (rp_stem_set_weights)."""
        conv = self.nnet.conv_seqs[0].conv
        return conv.weight.detach().contiguous(), conv.bias.detach().contiguous()

    def refresh_fused(self):
        """Rebuilds the small-image convolution matrices after a weight update (in place: captured graphs stay valid)."""
        if self.device.type == "cuda":
            self.nnet.refresh_dense()

    def predict_from_stem(self, stem, stem_relu=None, ops=None, logits=False):
        """stem: float32 [B, 16, (H+1)//2, (W+1)//2] from rp_leaf_stem -> (pi [B, A], v [B]) like predict_batch.
        With `ops` (the engine) the element-wise work runs through the engine's fused kernels; `stem_relu` (= relu(stem), which
        rp_leaf_stem can write alongside) is only read when stage 0 runs on the library path -- the stage kernel takes `stem` alone."""
        self.nnet.eval()
        with torch.no_grad():
            if ops is not None:
                pi, v = self.nnet.forward_from_stem_fused(stem, stem_relu, ops, logits=logits)  # logits: raw policy outputs for rp_commit_eval_logits
                return pi.contiguous(), v.reshape(-1).contiguous()
            if logits:
                raise ValueError("raw logits are only returned by the fused path (ops = the engine)")
            log_pi, v = self.nnet.forward_from_stem(stem)
            return torch.exp(log_pi).contiguous(), v.reshape(-1).contiguous()

    # ---- training ----------------------------------------------------------------------------
    def loss_pi(self, targets, outputs):
        return -torch.sum(targets * outputs) / targets.size()[0]

    def loss_v(self, targets, outputs):
        return torch.sum((targets - outputs.view(-1)) ** 2) / targets.size()[0]

    def train(self, examples):
        """examples: list of (state, pi, v) as CoachBPP hands over (reference :27-67)."""
        boards, pis, vs = zip(*examples)
        planes = torch.as_tensor(np.array(boards).astype(np.float64), dtype=torch.float32)
        target_pi = torch.as_tensor(np.array(pis), dtype=torch.float32)
        target_v = torch.as_tensor(np.array(vs).astype(np.float64), dtype=torch.float32)
        return self.train_tensors(planes.to(self.device), target_pi.to(self.device), target_v.to(self.device))

    def train_tensors(self, planes, target_pi, target_v):
        """Training on dense replay tensors (the schedule is `_train_loop`'s)."""
        return self._train_loop(planes.shape[0], lambda ids: (planes.index_select(0, ids), target_pi.index_select(0, ids), target_v.index_select(0, ids)),
                                planes.device)

    def train_packed(self, replay):
        """Training on a PackedReplay (replay.py): every minibatch is expanded to planes / pi / value by the engine's kernel
        (rp_expand_examples) right before its forward pass, so the replay set stays at ~0.4 KB per example instead of 55 KB."""
        hist = self._train_loop(len(replay), replay.expand, replay.device)
        replay.check()  # an index outside the packed arrays is reported by the expand kernel, not silently zero-filled
        return hist

    def _train_loop(self, n, fetch, device):
        """Same schedule as the reference: a fresh Adam with default hyper-parameters (`args.lr` is never read there,
        reference :31), `epochs` x floor(len / batch_size) steps, each on `batch_size` examples drawn WITH replacement
        from NumPy's global stream (:39-43), loss = -sum(pi * log p)/B + sum((v - v_hat)^2)/B (:87-91).
        fetch(ids) -> (planes, pi, value) of the examples `ids` (int64 device tensor).

        Data parallel (grad_hook set by distributed.attach): the step is STILL one batch of `batch_size` examples.  Rank 0 draws
        one seed from its global stream, every rank derives the same index stream from it and takes every world-th index of
        each batch; its loss terms are divided by the full batch size and the hook SUMS the gradients, so the update equals the
        single-process one on the same indices (up to float32 summation order) at the reference's learning rate."""
        optimizer = optim.Adam(self.nnet.parameters())
        history = []
        world, rank = 1, 0
        draw = np.random.randint
        if self.grad_hook is not None and torch.distributed.is_available() and torch.distributed.is_initialized():
            world, rank = torch.distributed.get_world_size(), torch.distributed.get_rank()
        if world > 1 or (self.grad_hook is not None and os.environ.get("RP_DIST_FORCE") == "1"):
            seed = torch.tensor([np.random.randint(1 << 31) if rank == 0 else 0], dtype=torch.int64, device=device)
            torch.distributed.broadcast(seed, src=0)
            draw = np.random.RandomState(int(seed.item())).randint
        B = int(self.args.batch_size)
        self.last_train_steps = 0
        for epoch in range(self.args.epochs):
            self.nnet.train()
            pi_losses, v_losses = AverageMeter(), AverageMeter()
            step_losses = []
            steps = int(n / B)
            cap = getattr(self.args, "max_train_steps_per_epoch", None)  # measurement aid (bench.py --coach-iter); the reference has no cap
            if cap:
                steps = min(steps, int(cap))
            timing = getattr(self, "step_timing", None)  # a list: (start event, end event) per optimiser step (bench.py --coach-iter)
            for _ in range(steps):
                if timing is not None and device.type == "cuda":
                    ev0 = torch.cuda.Event(enable_timing=True); ev0.record()
                ids_all = draw(n, size=B)
                ids = torch.as_tensor(ids_all[rank::world], device=device)
                x, t_pi, t_v = fetch(ids)
                out_pi, out_v = self.nnet(x)
                # loss_pi / loss_v (:87-91) with the FULL batch size as the divisor: a rank's slice contributes its share
                l_pi = -torch.sum(t_pi * out_pi) / B
                l_v = torch.sum((t_v - out_v.view(-1)) ** 2) / B
                optimizer.zero_grad()
                (l_pi + l_v).backward()
                l_pi, l_v = l_pi.detach(), l_v.detach()
                if self.grad_hook is not None:
                    l_pi, l_v = self.grad_hook(self.nnet, (l_pi, l_v))
                optimizer.step()
                if timing is not None and device.type == "cuda":
                    ev1 = torch.cuda.Event(enable_timing=True); ev1.record()
                    timing.append((ev0, ev1))
                step_losses.append(torch.stack([l_pi.reshape(()), l_v.reshape(())]))  # read back once per epoch: no host sync per step
                self.last_train_steps += 1
            if step_losses:
                for lp, lv in torch.stack(step_losses).cpu().tolist():
                    pi_losses.update(lp, B)
                    v_losses.update(lv, B)
            history.append((pi_losses.avg, v_losses.avg))
            if self.log:
                print("EPOCH ::: %d  Loss_pi=%s Loss_v=%s" % (epoch + 1, pi_losses, v_losses))
        return history

    # ---- checkpoints: {'state_dict': ...} exactly as the reference writes them (reference :93-111) ---------------
    def save_checkpoint(self, folder="checkpoint", filename="checkpoint.pth.tar"):
        os.makedirs(folder, exist_ok=True)
        torch.save({"state_dict": self.nnet.state_dict()}, os.path.join(folder, filename))

    def load_checkpoint(self, folder="checkpoint", filename="checkpoint.pth.tar"):
        path = os.path.join(folder, filename)
        if not os.path.exists(path):
            raise FileNotFoundError("No model in path {}".format(path))
        checkpoint = torch.load(path, map_location=self.device, weights_only=True)
        self.nnet.load_state_dict(checkpoint["state_dict"])
