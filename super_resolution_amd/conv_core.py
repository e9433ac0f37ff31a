"""One 3x3 convolution of the HIP conv core as an autograd building block.

The discriminator and the VGG19 feature extractor of the reference
(/root/reference/esrgan/models/esrgan_model.py:64-103) are stacks of 3x3 convolutions with other
channel widths, strides and epilogues than the generator's.  They reuse the generator's kernels through
this Function: activations stay in the blocked bf16 layout between layers, the forward is
``sr355::conv3x3_fwd`` (bias and LeakyReLU/ReLU fused), the data gradient is the same kernel on the
transposed weights and the weight/bias gradient is ``sr355::conv3x3_wgrad`` in blocks of 64 output
channels.  A stride-2 convolution is the stride-1 result sampled at even positions (same arithmetic as
``nn.Conv2d(k=3, s=2, p=1)``, four times the work); BatchNorm and pooling run as torch ops on views of the
blocked tensors.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor

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
    def backward(ctx, g: Tensor):
        return ops.from_blocked_now(g.contiguous(), 0, ctx.channels)


class _FromBlocked(torch.autograd.Function):
    @staticmethod
    def forward(ctx, buf: Tensor, channels: int) -> Tensor:
        ctx.units = buf.shape[2]
        return ops.from_blocked_now(buf.detach().contiguous(), 0, channels)

    @staticmethod
    def backward(ctx, g: Tensor):
        return ops.to_blocked_now(g.float().contiguous(), ctx.units, DTYPE_BF16), None


def to_blocked(x: Tensor) -> Tensor:
    """NCHW fp32 -> blocked bf16 [N, H, units, W, 8] (differentiable)."""
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

    def _sync(self, weight: Tensor, bias: Tensor) -> None:
        key = (weight.data_ptr(), weight._version, bias.data_ptr(), bias._version, weight.device)
        if key != self.key:
            self.key, self.fwd, self.bwd = key, None, None

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


def _packs_of(weight: Tensor) -> _PackedWeights:
    """The pack cache travels with the parameter object."""
    p = getattr(weight, "_sr355_packs", None)
    if p is None:
        p = _PackedWeights()
        weight._sr355_packs = p
    return p


class _Conv3x3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, weight: Tensor, bias: Tensor, slope: float, out_kind: str) -> Tensor:
        if x.dtype != torch.bfloat16 or x.dim() != 5 or not x.is_contiguous():
            raise RuntimeError("conv3x3: expected a contiguous blocked bf16 tensor [N, H, units, W, 8]")
        cout, cin = int(weight.shape[0]), int(weight.shape[1])
        N, H, U, W, _ = x.shape
        if U < units_for(cin):
            raise RuntimeError(f"conv3x3: input has {U} units, the weight needs {units_for(cin)}")
        wp, bp = _packs_of(weight).forward_pack(weight, bias)
        nchw_out = out_kind == "nchw"
        if out_kind not in ("blocked", "nchw", "f32"):
            raise RuntimeError(f"conv3x3: unknown output kind {out_kind!r}")
        if nchw_out:
            y = torch.empty((N, cout, H, W), dtype=torch.float32, device=x.device)
            ops.conv3x3_fwd_now(x, units_for(cin), wp, bp, cout, float(slope), 0, None, 0.0, None, 0.0, None, 0, None, y)
        elif out_kind == "f32":
            if cout % 16 or slope != 1.0:
                raise RuntimeError("conv3x3: fp32 outputs need a multiple of 16 channels and no fused activation")
            y = torch.empty((N, H, cout // 4, W, 4), dtype=torch.float32, device=x.device)
            ops.conv3x3_fwd_now(x, units_for(cin), wp, bp, cout, 1.0, 0, None, 0.0, None, 0.0, None, 0, y, None)
        else:
            if cout % 4:
                raise RuntimeError("conv3x3: blocked outputs need a multiple of 4 output channels")
            # the kernel writes every real channel of every pixel: only padding channels need the zero fill
            alloc = torch.empty if units_for(cout) * 8 == cout else torch.zeros
            y = alloc((N, H, units_for(cout), W, 8), dtype=torch.bfloat16, device=x.device)
            ops.conv3x3_fwd_now(x, units_for(cin), wp, bp, cout, float(slope), 0, None, 0.0, None, 0.0, y, 0, None, None)
        ctx.save_for_backward(x, weight, bias, y if (slope != 1.0 and out_kind == "blocked") else None)
        ctx.slope, ctx.out_kind = float(slope), out_kind
        return y

    @staticmethod
    def backward(ctx, dy: Tensor):
        x, weight, bias, y = ctx.saved_tensors
        cout, cin = int(weight.shape[0]), int(weight.shape[1])
        N, H, U, W, _ = x.shape
        if ctx.out_kind == "nchw":
            if ctx.slope != 1.0:
                raise RuntimeError("conv3x3: an NCHW output carries no fused activation")
            dyb = ops.to_blocked_now(dy.float().contiguous(), units_for(cout), DTYPE_BF16)
        elif ctx.out_kind == "f32":                 # [N,H,C/4,W,4] fp32 -> [N,H,C/8,W,8] bf16
            dyb = dy.reshape(N, H, cout // 8, 2, W, 4).permute(0, 1, 2, 4, 3, 5).reshape(N, H, cout // 8, W, 8)
            dyb = dyb.to(torch.bfloat16).contiguous()
        else:
            dyb = dy.contiguous()
            if y is not None:                       # derivative of the fused LeakyReLU / ReLU, one kernel
                # (the output has the sign of the pre-activation, so it can stand in for it)
                dyb = (torch.ops.aten.threshold_backward(dyb, y, 0.0) if ctx.slope == 0.0
                       else torch.ops.aten.leaky_relu_backward(dyb, y, ctx.slope, False))
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wt, zero_bias = _packs_of(weight).backward_pack(weight, bias)
            n_out = (cin + 3) // 4 * 4
            alloc = torch.empty if U * 8 == n_out else torch.zeros
            dx = alloc((N, H, U, W, 8), dtype=torch.bfloat16, device=x.device)
            ops.conv3x3_fwd_now(dyb, units_for(cout), wt, zero_bias, n_out, 1.0, 0, None, 0.0, None, 0.0, dx, 0, None, None)
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dw = torch.empty_like(weight, dtype=torch.float32)
            db = torch.empty((cout,), dtype=torch.float32, device=x.device)
            ops.conv3x3_wgrad_all(x, cin, dyb, cout, dw, db)     # 64 output channels a kernel call, looped inside
        return dx, dw, db, None, None


def conv3x3(x: Tensor, weight: Tensor, bias: Tensor, slope: float = 1.0, nchw_out: bool = False,
            f32_out: bool = False) -> Tensor:
    """3x3 / stride 1 / pad 1 convolution of a blocked bf16 tensor with an OIHW fp32 weight.  ``slope`` is the
    fused LeakyReLU slope (1.0 none, 0.0 ReLU).  The result is blocked bf16 [N,H,units,W,8]; ``nchw_out`` returns
    NCHW fp32 instead, ``f32_out`` the fp32 layout [N,H,C/4,W,4] (what a following normalisation should read:
    subtracting a batch mean from bf16-rounded values amplifies their rounding)."""
    return _Conv3x3.apply(x, weight, bias, slope, "nchw" if nchw_out else ("f32" if f32_out else "blocked"))


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
    window in row-major order (bf16 activations tie often)."""

    @staticmethod
    def forward(ctx, x: Tensor) -> Tensor:
        H2, W2 = x.shape[1] // 2 * 2, x.shape[3] // 2 * 2
        a, b = x[:, 0:H2:2, :, 0:W2:2], x[:, 0:H2:2, :, 1:W2:2]
        c, d = x[:, 1:H2:2, :, 0:W2:2], x[:, 1:H2:2, :, 1:W2:2]
        m = torch.maximum(torch.maximum(a, b), torch.maximum(c, d))
        ia = a == m
        ib = (b == m) & ~ia
        ic = (c == m) & ~ia & ~ib
        ctx.save_for_backward(ia, ib, ic)
        ctx.shape = x.shape
        return m.contiguous()

    @staticmethod
    def backward(ctx, g: Tensor):
        ia, ib, ic = ctx.saved_tensors
        H2, W2 = ctx.shape[1] // 2 * 2, ctx.shape[3] // 2 * 2
        dx = g.new_zeros(ctx.shape)
        zero = g.new_zeros(())
        dx[:, 0:H2:2, :, 0:W2:2] = torch.where(ia, g, zero)
        dx[:, 0:H2:2, :, 1:W2:2] = torch.where(ib, g, zero)
        dx[:, 1:H2:2, :, 0:W2:2] = torch.where(ic, g, zero)
        dx[:, 1:H2:2, :, 1:W2:2] = torch.where(ia | ib | ic, zero, g)
        return dx


def max_pool2x2(x: Tensor) -> Tensor:
    """2x2 / stride 2 max pooling of a blocked tensor (odd trailing rows / columns dropped, as nn.MaxPool2d)."""
    return _MaxPool2x2.apply(x)
