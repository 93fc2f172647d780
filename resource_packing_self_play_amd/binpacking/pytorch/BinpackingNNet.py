"""Policy/value network of the bin-packing evaluator (IMPALA-style CNN).

Architecture and parameter names follow the reference so that its checkpoints
(`{'state_dict': ...}` with `conv_seqs.{0,1,2}.{conv,res_block{0,1}.conv{0,1}}`,
`hidden_fc`, `logits_fc`, `value_fc`) load unchanged:
reference xw_mcts/binpacking/pytorch/BinpackingNNet.py:15-81.

    3 x [conv3x3(pad 1) -> maxpool(3, stride 2, pad 1) -> 2 x residual(relu,conv,relu,conv,+skip)]
    with 16/32/32 channels -> flatten -> relu -> fc 256 -> relu -> {fc A -> log_softmax, fc 1 -> tanh}

The module is FP32 only (the parity bar is 1e-5 against the CPU reference); it
is the body that `NNetWrapper` and the batched evaluator run on PyTorch-ROCm.
"""
import torch
from torch import nn
from torch.nn import functional as F

STAGE_CHANNELS = (16, 32, 32)
HIDDEN = 256


def _conv3x3(cin, cout):
    return nn.Conv2d(cin, cout, kernel_size=3, padding=1)


class _Residual(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv0 = _conv3x3(ch, ch)
        self.conv1 = _conv3x3(ch, ch)

    def forward(self, x):
        y = self.conv0(F.relu(x))
        y = self.conv1(F.relu(y))
        return y + x


class _Stage(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = _conv3x3(cin, cout)
        self.res_block0 = _Residual(cout)
        self.res_block1 = _Residual(cout)

    def forward(self, x):
        x = F.max_pool2d(self.conv(x), kernel_size=3, stride=2, padding=1)
        return self.res_block1(self.res_block0(x))


def stage_shapes(in_planes, board_h, board_w):
    """[(C, H, W)] after each stage; spatial size halves rounding up."""
    out, h, w = [], board_h, board_w
    for ch in STAGE_CHANNELS:
        h, w = (h + 1) // 2, (w + 1) // 2
        out.append((ch, h, w))
    return out


class BinPackingNNet(nn.Module):
    def __init__(self, game, args):
        super().__init__()
        self.board_h, self.board_w = game.getBoardSize()
        self.action_size = game.getActionSize()
        self.args = args
        self.in_channels = int(args.num_items) + int(args.num_bins)
        shapes = stage_shapes(self.in_channels, self.board_h, self.board_w)
        cins = (self.in_channels,) + STAGE_CHANNELS[:-1]
        self.conv_seqs = nn.ModuleList(_Stage(ci, co) for ci, co in zip(cins, STAGE_CHANNELS))
        c, h, w = shapes[-1]
        self.hidden_fc = nn.Linear(c * h * w, HIDDEN)
        self.logits_fc = nn.Linear(HIDDEN, self.action_size)
        self.value_fc = nn.Linear(HIDDEN, 1)

    def trunk(self, x):
        for stage in self.conv_seqs:
            x = stage(x)
        x = F.relu(torch.flatten(x, start_dim=1))
        return F.relu(self.hidden_fc(x))

    def heads(self, z):
        return F.log_softmax(self.logits_fc(z), dim=1), torch.tanh(self.value_fc(z))

    def forward(self, x):
        return self.heads(self.trunk(x))

    # ---- small-image convolutions as matrices --------------------------------------------------------------------------
    # On a 3x3 (or smaller) image a padded 3x3 convolution is a dense (C*h*w) x (C*h*w) matrix; MIOpen still pays a full
    # Winograd launch for it.  The matrices are read off the convolution itself (its response to the identity batch, so every
    # entry is exactly one weight) and refreshed in place after weight updates, which keeps captured HIP graphs valid.
    # Both memory orders are kept: NCHW-flattened (c, h, w) and channels-last-flattened (h, w, c).
    DENSE_MAX_PIXELS = 9
    # image sizes the engine's stage kernels take: one wave per group of leaves up to 128 / 80 pixels, one workgroup per group of leaves
    # above that (rp_nn_resstage16 / rp_nn_resstage32 / rp_nn_convpool32: 25x25x16 and 13x13x32 at the 50x50 board)
    STAGE16_MAX_PIXELS = 640
    STAGE32_MAX_PIXELS = 512
    use_resblock_kernel = True
    fused_linear_relu = hasattr(torch, "_addmm_activation")

    def _set_cached(self, key, value):
        if not hasattr(self, "_dense"):
            self._dense = {}
        if key in self._dense and self._dense[key].shape == value.shape:
            self._dense[key].copy_(value)
        else:
            self._dense[key] = value.contiguous()

    def refresh_dense(self):
        shapes = stage_shapes(self.in_channels, self.board_h, self.board_w)
        with torch.no_grad():
            for si, stage in enumerate(self.conv_seqs):
                ch, h, w = shapes[si]
                if h * w > self.DENSE_MAX_PIXELS:
                    continue
                n = ch * h * w
                for bi, blk in enumerate((stage.res_block0, stage.res_block1)):
                    for ci, conv in enumerate((blk.conv0, blk.conv1)):
                        eye = torch.eye(n, device=conv.weight.device, dtype=conv.weight.dtype)
                        key = "%d:b%dc%d" % (si, bi, ci)
                        self._set_cached(key, F.conv2d(eye.view(n, ch, h, w), conv.weight, None, padding=1).flatten(1))
                        resp = F.conv2d(eye.view(n, h, w, ch).permute(0, 3, 1, 2), conv.weight, None, padding=1)
                        self._set_cached(key + ":cl", resp.permute(0, 2, 3, 1).reshape(n, n))
            ch, h, w = shapes[-1]  # hidden_fc on channels-last features: columns reordered from (c, h, w) to (h, w, c)
            self._set_cached("hidden:cl", self.hidden_fc.weight.view(-1, ch, h, w).permute(0, 2, 3, 1).reshape(self.hidden_fc.out_features, -1))
        return self._dense

    def refresh_frags(self, ops):
        """MFMA B-fragment copies of the residual convolutions for the engine's fused stage kernels (rp_nn_resstage16 for
        16-channel stages on <= 640-pixel images, rp_nn_resstage32 for 32-channel stages on <= 512-pixel images; rp_nn_resblock16
        takes single 16-channel blocks); refreshed in place after weight updates.  The four fragments of a stage are slices
        of one buffer, next to a [4][C] copy of the biases, both in execution order."""
        if not hasattr(self, "_dense"):
            self._dense = {}
        keep = []
        shapes = stage_shapes(self.in_channels, self.board_h, self.board_w)
        with torch.no_grad():
            for si, stage in enumerate(self.conv_seqs):
                ch, h, w = shapes[si]
                if not ((ch == 16 and h * w <= self.STAGE16_MAX_PIXELS) or (ch == 32 and h * w <= self.STAGE32_MAX_PIXELS)):
                    continue
                n = 36 * 64 * (ch // 16) ** 2
                skey, bkey = "stagefrag:%d" % si, "stagebias:%d" % si
                if skey not in self._dense:
                    dev = stage.conv.weight.device
                    self._dense[skey] = torch.empty(4 * n, device=dev, dtype=torch.float32)
                    self._dense[bkey] = torch.empty(4 * ch, device=dev, dtype=torch.float32)
                k = 0
                for bi, blk in enumerate((stage.res_block0, stage.res_block1)):
                    for ci, conv in enumerate((blk.conv0, blk.conv1)):
                        key = "frag:%d:b%dc%d" % (si, bi, ci)
                        self._dense[key] = self._dense[skey][k * n:(k + 1) * n]
                        self._dense[bkey][k * ch:(k + 1) * ch].copy_(conv.bias.detach())
                        wt = conv.weight.detach().contiguous()  # plain [C][C][3][3] order whatever the parameter's format
                        keep.append(wt)
                        (ops.nn_pack_conv16 if ch == 16 else ops.nn_pack_conv32)(wt, self._dense[key])
                        k += 1
            for si, stage in enumerate(self.conv_seqs):  # first convolution (+ max-pool) of the 32-channel stages: rp_nn_convpool32
                if si == 0 or shapes[si][0] != 32:
                    continue
                cin, h, w = shapes[si - 1]
                if not ((cin == 16 and h * w <= self.STAGE16_MAX_PIXELS) or (cin == 32 and h * w <= self.STAGE32_MAX_PIXELS)):
                    continue
                key = "entryfrag:%d" % si
                if key not in self._dense:
                    self._dense[key] = torch.empty(9 * cin * 32, device=stage.conv.weight.device, dtype=torch.float32)
                wt = stage.conv.weight.detach().contiguous()
                keep.append(wt)
                ops.nn_pack_conv32(wt, self._dense[key])
        return keep

    @staticmethod
    def _is_cl(x):
        return x.dim() == 4 and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last)

    def _conv_nobias(self, x, conv, si, which):
        cl = self._is_cl(x)
        mt = getattr(self, "_dense", {}).get("%d:%s%s" % (si, which, ":cl" if cl else ""))
        if mt is None:
            return F.conv2d(x, conv.weight, None, padding=1)
        b, _, h, w = x.shape
        if cl:
            return torch.mm(x.permute(0, 2, 3, 1).reshape(b, -1), mt).view(b, h, w, conv.out_channels).permute(0, 3, 1, 2)
        return torch.mm(x.flatten(1), mt).view(b, conv.out_channels, h, w)

    def forward_from_stem_fused(self, y, y_relu, ops, logits=False):
        """Same network as forward_from_stem with the element-wise work fused: convolutions run without bias through
        PyTorch-ROCm, and `ops` (an engine: rp_nn_bias_relu / rp_nn_bias_residual / rp_nn_bias_pool) applies bias + ReLU,
        bias + skip (+ the next block's ReLU) and bias + max-pool in one pass each -- 4 kernels per residual block
        instead of 7, the same float32 operations in the same order.  Convolutions on <= 3x3 images run as one GEMM each
        when refresh_dense() has been called.  Works on NCHW-contiguous or channels-last tensors (MIOpen's FP32 kernels
        are 20-30 % faster on the latter).  y_relu = relu(y) or None.  Returns (softmax, tanh) -- or (raw logits, tanh) with
        `logits=True`, for rp_commit_eval_logits, which takes the softmax inside the commit kernel."""
        cl = self._is_cl(y)
        fmt = torch.channels_last if cl else torch.contiguous_format
        x, xr = y, y_relu
        for si, stage in enumerate(self.conv_seqs):
            ef = getattr(self, "_dense", {}).get("entryfrag:%d" % si) if cl and self.use_resblock_kernel and si > 0 else None
            if ef is not None:  # convolution + bias + max-pool in one kernel on the FP32 matrix cores
                b, _, h, w = x.shape
                xin, x = x, torch.empty((b, stage.conv.out_channels, (h + 1) // 2, (w + 1) // 2), device=x.device, dtype=x.dtype, memory_format=fmt)
                ops.nn_convpool32(xin, ef, stage.conv.bias, x)
                xr = None
            elif si > 0:
                c = F.conv2d(x, stage.conv.weight, None, padding=1)
                b, ch, h, w = c.shape
                x = torch.empty((b, ch, (h + 1) // 2, (w + 1) // 2), device=c.device, dtype=c.dtype, memory_format=fmt)
                xr = torch.empty_like(x)
                ops.nn_bias_pool(c, stage.conv.bias, x, xr)
            sf = getattr(self, "_dense", {}).get("stagefrag:%d" % si) if cl and self.use_resblock_kernel else None
            if sf is not None:  # both blocks of the stage in one kernel on the FP32 matrix cores
                last = si == len(self.conv_seqs) - 1
                out = torch.empty_like(x)
                out_r = torch.empty_like(x) if last else None  # only the flatten -> ReLU -> hidden_fc path reads relu(out)
                (ops.nn_resstage16 if x.shape[1] == 16 else ops.nn_resstage32)(x, sf, self._dense["stagebias:%d" % si], out, out_r)
                x, xr = out, out_r
                continue
            if xr is None:
                xr = torch.relu(x)
            for bi, blk in enumerate((stage.res_block0, stage.res_block1)):
                f0 = getattr(self, "_dense", {}).get("frag:%d:b%dc0" % (si, bi)) if cl and self.use_resblock_kernel and x.shape[1] == 16 else None
                if f0 is not None:  # whole block in one kernel on the FP32 matrix cores
                    out, out_r = torch.empty_like(x), torch.empty_like(x)
                    ops.nn_resblock16(x, f0, blk.conv0.bias, self._dense["frag:%d:b%dc1" % (si, bi)], blk.conv1.bias, out, out_r)
                    x, xr = out, out_r
                    continue
                c0 = self._conv_nobias(xr, blk.conv0, si, "b%dc0" % bi)
                ops.nn_bias_relu(c0, blk.conv0.bias)
                c1 = self._conv_nobias(c0, blk.conv1, si, "b%dc1" % bi)
                out, out_r = torch.empty_like(c1), torch.empty_like(c1)
                ops.nn_bias_residual(c1, blk.conv1.bias, x, out, out_r)
                x, xr = out, out_r
        hw = getattr(self, "_dense", {}).get("hidden:cl") if cl else None
        feats, wmat = (xr.permute(0, 2, 3, 1).reshape(xr.shape[0], -1), hw) if cl and hw is not None else (torch.flatten(xr, start_dim=1), self.hidden_fc.weight)
        if self.fused_linear_relu:  # bias + ReLU as the GEMM's epilogue (hipBLASLt) instead of a second pass over z
            z = torch._addmm_activation(self.hidden_fc.bias, feats, wmat.t())
        else:
            z = F.linear(feats, wmat, None)
            ops.nn_bias_relu(z.view(z.shape[0], z.shape[1], 1, 1), self.hidden_fc.bias)
        if hasattr(ops, "nn_value_head") and z.shape[1] % 4 == 0:  # value_fc + tanh in one pass over z instead of a 1-column GEMM + 2 kernels
            v = torch.empty((z.shape[0], 1), device=z.device, dtype=z.dtype)
            ops.nn_value_head(z, self.value_fc.weight, self.value_fc.bias, v)
        else:
            v = torch.tanh(self.value_fc(z))
        out = self.logits_fc(z)
        return (out if logits else torch.softmax(out, dim=1)), v

    def forward_from_stem(self, y):
        """y = max_pool2d(conv_seqs[0].conv(x), 3, 2, 1), e.g. from the engine's rp_leaf_stem: the rest of the network."""
        st0 = self.conv_seqs[0]
        y = st0.res_block1(st0.res_block0(y))
        for stage in self.conv_seqs[1:]:
            y = stage(y)
        y = F.relu(torch.flatten(y, start_dim=1))
        return self.heads(F.relu(self.hidden_fc(y)))
