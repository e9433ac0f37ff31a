"""One 3x3 convolution of the HIP conv core as an autograd building block.

The discriminator and the VGG19 feature extractor of the reference
(/root/reference/esrgan/models/esrgan_model.py:64-103) are stacks of 3x3 convolutions with other
channel widths, strides and epilogues than the generator's.  They reuse the generator's kernels through
this Function: activations stay in the blocked bf16 layout between layers, the forward is
``sr355::conv3x3_fwd`` (bias and LeakyReLU/ReLU fused), the data gradient is the same kernel on the
transposed weights and the weight/bias gradient is ``sr355::conv3x3_wgrad`` in blocks of 64 output
channels.  A stride-2 convolution (``nn.Conv2d(k=3, s=2, p=1)``) is computed at its output positions only; its data
and weight gradients read the output gradient as its zero-inserted x2 image through the same kernels.  BatchNorm +
LeakyReLU behind a conv (``conv_bn_act``), 2x2 max pooling (``max_pool2x2``) and the mean L1 / MSE losses
(``l1_loss`` / ``mse_loss``) are HIP kernels on the same layout, forward and backward.
"""
from __future__ import annotations

from typing import Optional, Tuple

import os

import torch
from torch import Tensor
from torch.autograd.function import once_differentiable

from . import ops
from ._lib import DTYPE_BF16


def units_for(channels: int) -> int:
    """Blocked units (8 bf16 channels each) holding ``channels``; conv stages consume units in pairs."""
    return ((channels + 7) // 8 + 1) // 2 * 2


def _pad32(c: int) -> int:
    return (c + 31) // 32 * 32


class _ToBlocked(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor) -> Tensor:
        ctx.channels = x.shape[1]
        return ops.to_blocked_now(x.detach().float().contiguous(), units_for(x.shape[1]), DTYPE_BF16)

    @staticmethod
    @once_differentiable
    def backward(ctx, g: Tensor):
        return ops.from_blocked_now(g.contiguous(), 0, ctx.channels)


class _FromBlocked(torch.autograd.Function):
    @staticmethod
    def forward(ctx, buf: Tensor, channels: int) -> Tensor:
        ctx.units = buf.shape[2]
        return ops.from_blocked_now(buf.detach().contiguous(), 0, channels)

    @staticmethod
    @once_differentiable
    def backward(ctx, g: Tensor):
        return ops.to_blocked_now(g.float().contiguous(), ctx.units, DTYPE_BF16), None


def to_blocked(x: Tensor) -> Tensor:
    """NCHW fp32 -> blocked bf16 [N, H, units, W, 8] (differentiable)."""
    if x.dim() != 4 or not x.is_floating_point():
        # (nn.Conv2d refuses integer images the same way)
        raise RuntimeError(f"conv core: expected a floating-point NCHW tensor; Input type ({x.dtype}, shape "
                           f"{tuple(x.shape)}) and weight type (torch.float32) should be the same")
    if 0 in x.shape:
        raise RuntimeError(f"conv core: a tensor without elements {tuple(x.shape)} (an empty batch or image) has no "
                           "blocked form; skip the call for an empty batch")
    return _ToBlocked.apply(x)


def from_blocked(buf: Tensor, channels: int) -> Tensor:
    """Blocked bf16 -> NCHW fp32 with ``channels`` channels (differentiable)."""
    return _FromBlocked.apply(buf, channels)


class _PackedWeights:
    """Forward / transposed packs of one conv, rebuilt when the parameter changes."""

    def __init__(self) -> None:
        self.key: Optional[Tuple] = None
        self.fwd: Optional[Tuple[Tensor, Tensor]] = None
        self.bwd: Optional[Tuple[Tensor, Tensor]] = None
        self.bwd_split: Optional[Tuple[Tensor, Tensor]] = None

    def _sync(self, weight: Tensor, bias: Tensor) -> None:
        key = (weight.data_ptr(), weight._version, bias.data_ptr(), bias._version, weight.device)
        if key != self.key:
            self.key, self.fwd, self.bwd, self.bwd_split = key, None, None, None

    def forward_pack(self, weight: Tensor, bias: Tensor) -> Tuple[Tensor, Tensor]:
        self._sync(weight, bias)
        if self.fwd is None:
            self.fwd = ops.pack_weight(weight.detach(), bias.detach(), DTYPE_BF16)
        return self.fwd

    def backward_pack(self, weight: Tensor, bias: Tensor) -> Tuple[Tensor, Tensor]:
        self._sync(weight, bias)
        if self.bwd is None:
            cout, cin = weight.shape[0], weight.shape[1]
            n_out = (cin + 3) // 4 * 4                     # blocked outputs come in channel quads
            units = units_for(cout)
            wt = torch.zeros(units * 9 * _pad32(n_out) * 16, dtype=torch.uint8, device=weight.device)
            ops.pack_weight_transposed(weight.detach(), 0, cin, 1.0, wt, units, 0, DTYPE_BF16)
            self.bwd = (wt, torch.zeros(_pad32(n_out), dtype=torch.float32, device=weight.device))
        return self.bwd


    def backward_pack_split(self, weight: Tensor, bias: Tensor) -> Tuple[Tensor, Tensor]:
        """The transposed pack for an output gradient carried as hi + lo bf16 pairs (channels [0, cout) the bf16
        value, [cout, 2 cout) its rounding residue): every output channel's weights appear twice."""
        self._sync(weight, bias)
        if self.bwd_split is None:
            cout, cin = weight.shape[0], weight.shape[1]
            n_out = (cin + 3) // 4 * 4
            units = units_for(2 * cout)
            wt = torch.zeros(units * 9 * _pad32(n_out) * 16, dtype=torch.uint8, device=weight.device)
            w2 = torch.cat([weight.detach(), weight.detach()], dim=0).contiguous()
            ops.pack_weight_transposed(w2, 0, cin, 1.0, wt, units, 0, DTYPE_BF16)
            self.bwd_split = (wt, torch.zeros(_pad32(n_out), dtype=torch.float32, device=weight.device))
        return self.bwd_split


def split_cotangent_fits(cout: int) -> bool:
    """An NCHW output gradient of so few channels that value + rounding residue fit the units of the value alone."""
    return 2 * cout <= units_for(cout) * 8


def _packs_of(weight: Tensor) -> _PackedWeights:
    """The pack cache travels with the parameter object."""
    p = getattr(weight, "_sr355_packs", None)
    if p is None:
        p = _PackedWeights()
        weight._sr355_packs = p
    return p


def prepare_packs(convs, backward: bool) -> None:
    """Brings the packs of several convs up to date in ONE launch per form (and one zero fill) instead of one launch, one
    zero fill and one allocation per conv: a network repacks all of its convs after every optimizer step.  ``convs`` are
    ``nn.Conv2d`` modules (3x3); ``backward`` also prepares the whole-conv data-gradient packs.  What is left to the
    per-conv path (a split-cotangent pack) is built there on first use, as before."""
    if os.environ.get("SR355_BATCH_PACKS", "1") == "0":      # (measurement knob: one launch per conv and form, as before)
        return
    stale_f, stale_b = [], []
    for c in convs:
        w, b = c.weight, c.bias
        if w.dtype != torch.float32 or b is None or not w.is_cuda:
            return                                   # the per-conv path reports what is wrong with such a module
        packs = _packs_of(w)
        packs._sync(w, b)
        if packs.fwd is None:
            stale_f.append((packs, w, b))
        if backward and packs.bwd is None:
            stale_b.append((packs, w))
    if len(stale_f) > 1:
        built = ops.pack_weights_batch([w.detach() for _p, w, _b in stale_f], [b.detach() for _p, _w, b in stale_f], DTYPE_BF16)
        for (packs, _w, _b), pair in zip(stale_f, built):
            packs.fwd = pair
    if len(stale_b) > 1:
        built = ops.pack_weights_transposed_batch([w.detach() for _p, w in stale_b], DTYPE_BF16)
        for (packs, _w), pair in zip(stale_b, built):
            packs.bwd = pair


def _conv_forward(x: Tensor, weight: Tensor, bias: Tensor, slope: float, out_kind: str, stride: int) -> Tensor:
    """One conv launch: blocked bf16 in, ``out_kind`` out ("blocked" bf16, "f32" [N,H,C/4,W,4], "nchw" fp32)."""
    if x.dtype != torch.bfloat16 or x.dim() != 5 or not x.is_contiguous():
        raise RuntimeError("conv3x3: expected a contiguous blocked bf16 tensor [N, H, units, W, 8]")
    if stride not in (1, 2):
        raise RuntimeError(f"conv3x3: stride {stride} is not supported (1 or 2)")
    cout, cin = int(weight.shape[0]), int(weight.shape[1])
    N, Hi, U, Wi, _ = x.shape
    H, W = ((Hi + 1) // 2, (Wi + 1) // 2) if stride == 2 else (Hi, Wi)
    if U < units_for(cin):
        raise RuntimeError(f"conv3x3: input has {U} units, the weight needs {units_for(cin)}")
    wp, bp = _packs_of(weight).forward_pack(weight, bias)
    out_act = out_f32 = out_nchw = None
    if out_kind == "nchw":
        y = out_nchw = torch.empty((N, cout, H, W), dtype=torch.float32, device=x.device)
    elif out_kind == "f32":
        if cout % 16 or slope != 1.0:
            raise RuntimeError("conv3x3: fp32 outputs need a multiple of 16 channels and no fused activation")
        y = out_f32 = torch.empty((N, H, cout // 4, W, 4), dtype=torch.float32, device=x.device)
    elif out_kind == "blocked":
        if cout % 4:
            raise RuntimeError("conv3x3: blocked outputs need a multiple of 4 output channels")
        # the kernel writes every real channel of every pixel: only padding channels need the zero fill
        alloc = torch.empty if units_for(cout) * 8 == cout else torch.zeros
        y = out_act = alloc((N, H, units_for(cout), W, 8), dtype=torch.bfloat16, device=x.device)
    else:
        raise RuntimeError(f"conv3x3: unknown output kind {out_kind!r}")
    if stride == 1:
        ops.conv3x3_fwd_now(x, units_for(cin), wp, bp, cout, float(slope), 0, None, 0.0, None, 0.0, out_act, 0, out_f32,
                            out_nchw)
    else:
        ops.launch_conv(ops.conv_args(x, units_for(cin), wp, bp, cout, float(slope), 0, None, 0.0, None, 0.0, out_act, 0,
                                      out_f32, out_nchw, stride=2))
    return y


def _conv_backward(x: Tensor, weight: Tensor, bias: Tensor, dyb: Tensor, stride: int, need_dx: bool, need_dw: bool,
                   dx_f32: bool = False, dy_split: bool = False, gate: Optional[Tensor] = None, gate_slope: float = 1.0):
    """Data / weight / bias gradients of one conv from the blocked bf16 gradient ``dyb`` of its (pre-activation)
    output.  Stride 2: the data gradient reads ``dyb`` as its zero-inserted x2 image, the weight gradient likewise.
    ``dx_f32``: the data gradient as an fp32 tensor ``[N, H, cin/4, W, 4]`` (cin a multiple of 16) instead of blocked
    bf16 — for a consumer that subtracts means from it (a BatchNorm's backward).  ``dy_split``: ``dyb`` carries the
    gradient as hi + lo bf16 pairs (``to_blocked(..., split=1)``: channels [cout, 2 cout) hold the rounding residue); the
    data gradient multiplies both, the weight gradient reads the hi part.  ``gate`` (stride 1, blocked data gradient): a
    blocked bf16 tensor of the data gradient's geometry whose sign gates it in the conv's epilogue — where the gate is not
    positive the gradient is multiplied by ``gate_slope`` (0: the ReLU that produced this conv's input, whose output is
    ``x`` itself)."""
    cout, cin = int(weight.shape[0]), int(weight.shape[1])
    N, H, U, W, _ = x.shape
    dx = dw = db = None
    if need_dx:
        packs = _packs_of(weight)
        wt, zero_bias = packs.backward_pack_split(weight, bias) if dy_split else packs.backward_pack(weight, bias)
        if dy_split:
            if stride != 1:
                raise RuntimeError("conv3x3: a split output gradient is for stride-1 convs")
            cout = 2 * cout          # (the data gradient's input channels)
        n_out = (cin + 3) // 4 * 4
        if dx_f32:
            if cin % 16:
                raise RuntimeError("conv3x3: an fp32 data gradient needs a multiple of 16 input channels")
            dx = torch.empty((N, H, cin // 4, W, 4), dtype=torch.float32, device=x.device)
            out_act, out_f32 = None, dx
        else:
            alloc = torch.empty if U * 8 == n_out else torch.zeros
            dx = alloc((N, H, U, W, 8), dtype=torch.bfloat16, device=x.device)
            out_act, out_f32 = dx, None
        if gate is not None and (stride != 1 or dx_f32 or gate.shape != dx.shape):
            raise RuntimeError("conv3x3: a gated data gradient is a stride-1 conv with a blocked result of the gate's shape")
        if stride == 1:
            ops.conv3x3_fwd_now(dyb, units_for(cout), wt, zero_bias, n_out, 1.0, 0, None, 0.0, None, 0.0, out_act, 0,
                                out_f32, None, 0, gate, 0, float(gate_slope))
        else:
            ops.launch_conv(ops.conv_args(dyb, units_for(cout), wt, zero_bias, n_out, 1.0, 1, None, 0.0, None, 0.0, out_act,
                                          0, out_f32, None, zero_insert=True))
    if need_dw:
        cout = int(weight.shape[0])
        dw = torch.empty_like(weight, dtype=torch.float32)
        db = torch.empty((cout,), dtype=torch.float32, device=x.device)
        ops.conv3x3_wgrad_all(x, cin, dyb, cout, dw, db, stride=stride)   # 64 output channels a kernel call, looped inside
    return dx, dw, db


class _Conv3x3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, weight: Tensor, bias: Tensor, slope: float, out_kind: str, stride: int) -> Tensor:
        y = _conv_forward(x, weight, bias, slope, out_kind, stride)
        ctx.save_for_backward(x, weight, bias, y if (slope != 1.0 and out_kind == "blocked") else None)
        ctx.slope, ctx.out_kind, ctx.stride = float(slope), out_kind, stride
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy: Tensor):
        x, weight, bias, y = ctx.saved_tensors
        cout = int(weight.shape[0])
        dy_split = False
        if ctx.out_kind == "nchw":
            if ctx.slope != 1.0:
                raise RuntimeError("conv3x3: an NCHW output carries no fused activation")
            # few output channels (a network's last conv): the cotangent as hi + lo bf16 pairs, so that what the layers
            # below subtract from each other is not its rounding noise
            dy_split = ctx.stride == 1 and split_cotangent_fits(cout)
            dyb = ops.to_blocked_now(dy.float().contiguous(), units_for(cout), DTYPE_BF16, 1 if dy_split else 0)
        elif ctx.out_kind == "f32":                 # [N,H,C/4,W,4] fp32 -> [N,H,C/8,W,8] bf16
            N, H, _q, W, _c = dy.shape
            dyb = dy.reshape(N, H, cout // 8, 2, W, 4).permute(0, 1, 2, 4, 3, 5).reshape(N, H, cout // 8, W, 8)
            dyb = dyb.to(torch.bfloat16).contiguous()
        else:
            dyb = dy.contiguous()
            if y is not None:                       # derivative of the fused LeakyReLU / ReLU, one kernel
                # (the output has the sign of the pre-activation, so it can stand in for it)
                dyb = (torch.ops.aten.threshold_backward(dyb, y, 0.0) if ctx.slope == 0.0
                       else torch.ops.aten.leaky_relu_backward(dyb, y, ctx.slope, False))
        dx, dw, db = _conv_backward(x, weight, bias, dyb, ctx.stride, ctx.needs_input_grad[0],
                                    ctx.needs_input_grad[1] or ctx.needs_input_grad[2], dy_split=dy_split)
        return dx, dw, db, None, None, None


def conv3x3(x: Tensor, weight: Tensor, bias: Tensor, slope: float = 1.0, nchw_out: bool = False,
            f32_out: bool = False, stride: int = 1) -> Tensor:
    """3x3 / pad 1 convolution (stride 1 or 2) of a blocked bf16 tensor with an OIHW fp32 weight.  ``slope`` is the
    fused LeakyReLU slope (1.0 none, 0.0 ReLU).  The result is blocked bf16 [N,H,units,W,8]; ``nchw_out`` returns
    NCHW fp32 instead, ``f32_out`` the fp32 layout [N,H,C/4,W,4] (what a following normalisation should read:
    subtracting a batch mean from bf16-rounded values amplifies their rounding).  Stride 2 is computed at the output
    positions only; its gradients read the output gradient as its zero-inserted x2 image."""
    return _Conv3x3.apply(x, weight, bias, slope, "nchw" if nchw_out else ("f32" if f32_out else "blocked"), stride)


class _ConvBnAct(torch.autograd.Function):
    """conv3x3 -> BatchNorm2d -> LeakyReLU as the discriminator's ``conv_block`` composes them
    (/root/reference/esrgan/models/esrgan_model.py:67-72): the conv's fp32 output goes straight into the statistics
    and normalisation kernels, which write the next layer's blocked bf16 input; the backward pass returns the blocked
    bf16 gradient the conv's data / weight gradient kernels read.  No layout shuffle on either side."""

    @staticmethod
    def forward(ctx, x: Tensor, weight: Tensor, bias: Tensor, gamma: Tensor, beta: Tensor, bn, slope: float,
                stride: int) -> Tensor:
        cout = int(weight.shape[0])
        if cout % 16:
            raise RuntimeError("conv_bn_act: the normalised layer needs a multiple of 16 channels")
        if bn.momentum is None or not bn.track_running_stats or not bn.affine:
            raise RuntimeError("conv_bn_act: BatchNorm2d with affine parameters, running statistics and a fixed momentum")
        y32 = _conv_forward(x, weight, bias, 1.0, "f32", stride)
        N, H, _q, W, _c = y32.shape
        training = bool(bn.training)
        mean = torch.empty(cout, dtype=torch.float32, device=x.device)
        rstd = torch.empty(cout, dtype=torch.float32, device=x.device)
        out = torch.empty((N, H, cout // 8, W, 8), dtype=torch.bfloat16, device=x.device)
        ops.bn_lrelu_fwd_now(y32, gamma.detach(), beta.detach(), float(bn.eps), float(slope), training, float(bn.momentum),
                             bn.running_mean, bn.running_var, bn.num_batches_tracked, mean, rstd, out, 0)
        ctx.save_for_backward(x, weight, bias, gamma, beta, y32, mean, rstd)
        ctx.slope, ctx.stride, ctx.training = float(slope), stride, training
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dy: Tensor):
        x, weight, bias, gamma, beta, y32, mean, rstd = ctx.saved_tensors
        cout = int(weight.shape[0])
        dyb = torch.empty_like(dy, memory_format=torch.contiguous_format)
        dgamma = torch.empty(cout, dtype=torch.float32, device=dy.device)
        dbeta = torch.empty(cout, dtype=torch.float32, device=dy.device)
        ops.bn_lrelu_bwd_now(y32, dy.contiguous(), gamma.detach(), beta.detach(), mean, rstd, ctx.slope, ctx.training,
                             dyb, dgamma, dbeta, 0, 0)
        dx, dw, db = _conv_backward(x, weight, bias, dyb, ctx.stride, ctx.needs_input_grad[0],
                                    ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        return dx, dw, db, dgamma, dbeta, None, None, None


def conv_bn_act(x: Tensor, conv, bn, slope: float) -> Tensor:
    """``LeakyReLU(slope)(bn(conv(x)))`` for a 3x3 / pad 1 ``nn.Conv2d`` (stride 1 or 2) and an ``nn.BatchNorm2d``,
    blocked bf16 in and out; the BatchNorm module's mode and running buffers behave as in stock PyTorch."""
    return _ConvBnAct.apply(x, conv.weight, conv.bias, bn.weight, bn.bias, bn, slope, int(conv.stride[0]))


def _blocked_to_f32_layout(t: Tensor) -> Tensor:
    """blocked bf16 [N, H, C/8, W, 8] -> fp32 layout [N, H, C/4, W, 4]."""
    N, H, U, W, _ = t.shape
    return t.reshape(N, H, U, W, 2, 4).permute(0, 1, 2, 4, 3, 5).reshape(N, H, 2 * U, W, 4).float()


def _f32_layout_to_blocked(t: Tensor) -> Tensor:
    """fp32 layout [N, H, C/4, W, 4] -> blocked bf16 [N, H, C/8, W, 8]."""
    N, H, Q, W, _ = t.shape
    return t.reshape(N, H, Q // 2, 2, W, 4).permute(0, 1, 2, 4, 3, 5).reshape(N, H, Q // 2, W, 8).to(torch.bfloat16).contiguous()


class _ConvAdd(torch.autograd.Function):
    """``trunk + conv3x3(x)`` as ONE launch: the conv's epilogue reads the fp32 trunk, adds, and writes the new fp32
    trunk together with its bf16 copy (what the next conv reads).  The residual step of
    /root/reference/CycleGAN/CycleGAN_w_v1/CycleGAN.py:97-100 without a separate add or layout pass."""

    @staticmethod
    def forward(ctx, x: Tensor, weight: Tensor, bias: Tensor, trunk: Tensor):
        cout, cin = int(weight.shape[0]), int(weight.shape[1])
        if x.dtype != torch.bfloat16 or x.dim() != 5 or not x.is_contiguous():
            raise RuntimeError("conv3x3_add: expected a contiguous blocked bf16 tensor [N, H, units, W, 8]")
        N, H, U, W, _ = x.shape
        if cout % 16 or trunk.dtype != torch.float32 or tuple(trunk.shape) != (N, H, cout // 4, W, 4):
            raise RuntimeError("conv3x3_add: the trunk must be fp32 [N, H, cout/4, W, 4] with cout a multiple of 16")
        if U < units_for(cin):
            raise RuntimeError(f"conv3x3_add: input has {U} units, the weight needs {units_for(cin)}")
        wp, bp = _packs_of(weight).forward_pack(weight, bias)
        new_trunk = torch.empty_like(trunk, memory_format=torch.contiguous_format)
        act = torch.empty((N, H, cout // 8, W, 8), dtype=torch.bfloat16, device=x.device)
        ops.conv3x3_fwd_now(x, units_for(cin), wp, bp, cout, 1.0, 0, trunk.contiguous(), 1.0, None, 0.0, act, 0, new_trunk,
                            None)
        ctx.save_for_backward(x, weight, bias)
        return new_trunk, act

    @staticmethod
    @once_differentiable
    def backward(ctx, g_trunk, g_act):
        x, weight, bias = ctx.saved_tensors
        total = g_trunk
        if g_act is not None:
            ga = _blocked_to_f32_layout(g_act)
            total = ga if total is None else total + ga
        dx, dw, db = _conv_backward(x, weight, bias, _f32_layout_to_blocked(total), 1, ctx.needs_input_grad[0],
                                    ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        return dx, dw, db, total


def conv3x3_add(x: Tensor, weight: Tensor, bias: Tensor, trunk: Tensor) -> Tuple[Tensor, Tensor]:
    """``trunk + conv3x3(x)`` in one launch: returns (new fp32 trunk ``[N, H, C/4, W, 4]``, its blocked bf16 copy)."""
    return _ConvAdd.apply(x, weight, bias, trunk)


class _PixelShuffle2(torch.autograd.Function):
    """``nn.PixelShuffle(2)`` on a blocked tensor, one HIP kernel each way (the gradient is the inverse permutation)."""

    @staticmethod
    def forward(ctx, x: Tensor) -> Tensor:
        return ops.pixel_shuffle2_now(x.contiguous(), False)

    @staticmethod
    @once_differentiable
    def backward(ctx, g: Tensor):
        return ops.pixel_shuffle2_now(g.contiguous(), True)


def pixel_shuffle2(x: Tensor) -> Tensor:
    """``nn.PixelShuffle(2)``: blocked ``[N, H, 4U, W, c]`` -> ``[N, 2H, U, 2W, c]`` (bf16 or fp32 units)."""
    return _PixelShuffle2.apply(x)


def f32_as_nchw(y: Tensor) -> Tensor:
    """View of an fp32 conv output [N, H, C/4, W, 4] as [N, C, H, W] (no copy)."""
    N, H, Q, W, c = y.shape
    return y.permute(0, 2, 4, 1, 3).reshape(N, Q * c, H, W)


def blocked_as_nchw(x: Tensor) -> Tensor:
    """View of a blocked tensor [N, H, U, W, 8] as [N, U*8, H, W] (no copy)."""
    N, H, U, W, c = x.shape
    return x.permute(0, 2, 4, 1, 3).reshape(N, U * c, H, W)


def nchw_as_blocked(v: Tensor) -> Tensor:
    """[N, U*8, H, W] -> contiguous blocked bf16 [N, H, U, W, 8]."""
    N, C, H, W = v.shape
    return v.reshape(N, C // 8, 8, H, W).permute(0, 3, 1, 4, 2).contiguous().to(torch.bfloat16)


class _MaxPool2x2(torch.autograd.Function):
    """nn.MaxPool2d(2, 2) on a blocked tensor with its tie rule: the whole gradient goes to the first maximum of a
    window in row-major order (bf16 activations tie often).  Both directions are one HIP kernel; the backward pass
    recomputes the argmax from the saved input."""

    @staticmethod
    def forward(ctx, x: Tensor) -> Tensor:
        x = x.contiguous()
        ctx.save_for_backward(x)
        return ops.maxpool2x2_fwd_now(x)

    @staticmethod
    @once_differentiable
    def backward(ctx, g: Tensor):
        (x,) = ctx.saved_tensors
        return ops.maxpool2x2_bwd_now(x, g.contiguous())


def max_pool2x2(x: Tensor) -> Tensor:
    """2x2 / stride 2 max pooling of a blocked tensor (odd trailing rows / columns dropped, as nn.MaxPool2d)."""
    return _MaxPool2x2.apply(x)


class _MeanLoss(torch.autograd.Function):
    """mean |a - b| (mode 0) / mean (a - b)^2 (mode 1) with the gradient factor written by the same kernel pass."""

    @staticmethod
    def forward(ctx, a: Tensor, b, b_const: float, mode: int) -> Tensor:
        need = ctx.needs_input_grad[0] or (isinstance(b, Tensor) and ctx.needs_input_grad[1])
        loss, grad = ops.loss_fwd_now(a, b, float(b_const), int(mode), bool(need))
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g: Tensor):
        (grad,) = ctx.saved_tensors
        ga = grad * g
        return (ga if ctx.needs_input_grad[0] else None, -ga if ctx.needs_input_grad[1] else None, None, None)


def _loss(a: Tensor, b, mode: int) -> Tensor:
    a32 = a.float().contiguous()
    if isinstance(b, Tensor):
        if b.shape != a.shape:
            raise RuntimeError(f"loss: shapes differ ({tuple(a.shape)} vs {tuple(b.shape)})")
        return _MeanLoss.apply(a32, b.float().contiguous(), 0.0, mode)
    return _MeanLoss.apply(a32, None, float(b), mode)


def l1_loss(a: Tensor, b) -> Tensor:
    """``nn.L1Loss()(a, b)`` (mean reduction) in one reduction kernel; ``b`` may be a Python number."""
    return _loss(a, b, 0)


def mse_loss(a: Tensor, b) -> Tensor:
    """``nn.MSELoss()(a, b)`` (mean reduction) in one reduction kernel; ``b`` may be a Python number (the ones / zeros
    targets of the adversarial terms, /root/reference/esrgan/train.py:201-204, :214)."""
    return _loss(a, b, 1)
