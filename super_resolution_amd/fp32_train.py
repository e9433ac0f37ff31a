"""Training on the exact-fp32 kernels: the recorded forward of ``GeneratorRRDB`` / ``RRDB`` / ``ResidualDenseBlock``
built with ``precision="fp32"``.

The bf16 training path is one autograd node per network with its own saved-activation layout; this one is the
reference's module graph (/root/reference/esrgan/models/esrgan_model.py:18-24, :35-39, :55-62) written conv by conv:
every 3x3 conv is one autograd node on blocked fp32 tensors ``[N, H, C/4, W, 4]`` whose forward is the fp32 MFMA conv
(``sr355::conv3x3_fwd``, bias / LeakyReLU / nearest x2 fused), whose data gradient is the same kernel on the transposed
pack and whose weight / bias gradient is ``sr355::conv3x3_wgrad_f32``; the channel concatenation of a dense block is a
``torch.cat`` along the unit dimension of that layout and the skips are elementwise torch ops, both differentiated by
autograd.  It is the accuracy path (gradients within 1e-4 of the reference's fp32 autograd), not the fast one.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor
from torch.autograd.function import once_differentiable

from . import ops
from ._lib import DTYPE_F32

_SLOPE = 0.2


def _units4(channels: int) -> int:
    """Blocked fp32 units (4 channels each) holding ``channels``; conv stages consume units in pairs."""
    return ((channels + 3) // 4 + 1) // 2 * 2


def _pad32(c: int) -> int:
    return (c + 31) // 32 * 32


class _ToBlockedF32(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor) -> Tensor:
        ctx.channels = x.shape[1]
        return ops.to_blocked_now(x.detach().float().contiguous(), _units4(x.shape[1]), DTYPE_F32)

    @staticmethod
    @once_differentiable
    def backward(ctx, g: Tensor):
        return ops.from_blocked_now(g.contiguous(), 0, ctx.channels)


class _FromBlockedF32(torch.autograd.Function):
    @staticmethod
    def forward(ctx, buf: Tensor, channels: int) -> Tensor:
        ctx.units = buf.shape[2]
        return ops.from_blocked_now(buf.detach().contiguous(), 0, channels)

    @staticmethod
    @once_differentiable
    def backward(ctx, g: Tensor):
        return ops.to_blocked_now(g.float().contiguous(), ctx.units, DTYPE_F32), None


class _ConvF32(torch.autograd.Function):
    """``act(conv3x3(up(x)))`` on blocked fp32 tensors; ``nchw_out`` returns NCHW fp32 (the network's last conv)."""

    @staticmethod
    def forward(ctx, x: Tensor, weight: Tensor, bias: Tensor, slope: float, upsample: int, nchw_out: bool) -> Tensor:
        if x.dtype != torch.float32 or x.dim() != 5 or x.shape[4] != 4:
            raise RuntimeError("fp32 conv: expected a blocked fp32 tensor [N, H, units, W, 4]")
        x = x.contiguous()
        cout, cin = int(weight.shape[0]), int(weight.shape[1])
        N, Hin, U, Win, _ = x.shape
        if U < _units4(cin):
            raise RuntimeError(f"fp32 conv: input has {U} units, the weight needs {_units4(cin)}")
        H, W = Hin << upsample, Win << upsample
        wp, bp = ops.pack_weight_now(weight.detach(), bias.detach(), DTYPE_F32)
        out_act = out_nchw = None
        if nchw_out:
            y = out_nchw = torch.empty((N, cout, H, W), dtype=torch.float32, device=x.device)
        else:
            if cout % 4:
                raise RuntimeError("fp32 conv: blocked outputs need a multiple of 4 output channels")
            alloc = torch.empty if _units4(cout) * 4 == cout else torch.zeros
            y = out_act = alloc((N, H, _units4(cout), W, 4), dtype=torch.float32, device=x.device)
        ops.conv3x3_fwd_now(x, _units4(cin), wp, bp, cout, float(slope), int(upsample), None, 0.0, None, 0.0, out_act, 0,
                            None, out_nchw)
        ctx.save_for_backward(x, weight, y if (slope != 1.0 and not nchw_out) else None)
        ctx.slope, ctx.upsample, ctx.nchw_out = float(slope), int(upsample), bool(nchw_out)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy: Tensor):
        x, weight, y = ctx.saved_tensors
        cout, cin = int(weight.shape[0]), int(weight.shape[1])
        N, Hin, U, Win, _ = x.shape
        ups = ctx.upsample
        H, W = Hin << ups, Win << ups
        dev = x.device
        if ctx.nchw_out:
            if ctx.slope != 1.0:
                raise RuntimeError("fp32 conv: an NCHW output carries no fused activation")
            dyb = ops.to_blocked_now(dy.float().contiguous(), _units4(cout), DTYPE_F32)
        else:
            dyb = dy.contiguous()
            if y is not None:        # the output has the sign of the pre-activation, so it can stand in for it
                dyb = torch.ops.aten.leaky_relu_backward(dyb, y, ctx.slope, False)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            n_out = (cin + 3) // 4 * 4
            units = _units4(cout)
            wt = torch.zeros(units * 9 * _pad32(n_out) * 16, dtype=torch.uint8, device=dev)
            ops.pack_weight_transposed_now(weight.detach(), 0, cin, 1.0, wt, units, 0, DTYPE_F32)
            zero_bias = torch.zeros(_pad32(n_out), dtype=torch.float32, device=dev)
            alloc = torch.empty if U * 4 == n_out else torch.zeros
            full = alloc((N, H, U, W, 4), dtype=torch.float32, device=dev)
            ops.conv3x3_fwd_now(dyb, units, wt, zero_bias, n_out, 1.0, 0, None, 0.0, None, 0.0, full, 0, None, None)
            if ups:                  # adjoint of the nearest x2 read: the 2x2 sum
                dx = torch.empty((N, Hin, U, Win, 4), dtype=torch.float32, device=dev)
                ops.pool2x2_sum_now(full, None, 0, 1.0, None, dx, 0)
            else:
                dx = full
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dw = torch.empty_like(weight, dtype=torch.float32)
            db = torch.empty((cout,), dtype=torch.float32, device=dev)
            ops.conv3x3_wgrad_f32_now(x, cin, ups, dyb, 0, cout, 1.0, 0, dw, db)
        return dx, dw, db, None, None, None


def _conv(x: Tensor, conv, slope: float = 1.0, upsample: int = 0, nchw_out: bool = False) -> Tensor:
    return _ConvF32.apply(x, conv.weight, conv.bias, slope, upsample, nchw_out)


def dense_block(block, x: Tensor) -> Tensor:
    """esrgan_model.py:18-24 on a blocked fp32 tensor."""
    feats = x
    for k in range(1, 5):
        feats = torch.cat([feats, _conv(feats, getattr(block, f"conv{k}"), _SLOPE)], dim=2)
    return _conv(feats, block.conv5) * block.res_scale + x


def rrdb(block, x: Tensor) -> Tensor:
    """esrgan_model.py:35-39 on a blocked fp32 tensor."""
    out = dense_block(block.rdb3, dense_block(block.rdb2, dense_block(block.rdb1, x)))
    return out * block.res_scale + x


def block_forward(module, x: Tensor, is_rrdb: bool) -> Tensor:
    """Recorded fp32 forward of a stand-alone ``ResidualDenseBlock`` / ``RRDB`` (NCHW in, NCHW out)."""
    xb = _ToBlockedF32.apply(x)
    out = rrdb(module, xb) if is_rrdb else dense_block(module, xb)
    return _FromBlockedF32.apply(out, module.in_channels)


def generator_forward(net, x: Tensor) -> Tensor:
    """Recorded fp32 forward of ``GeneratorRRDB`` (esrgan_model.py:55-62): NCHW fp32 in, NCHW fp32 x4 image out."""
    fea = _conv(_ToBlockedF32.apply(x), net.conv_first)
    body = fea
    for block in net.body:
        body = rrdb(block, body)
    fea = fea + _conv(body, net.trunk_conv)
    fea = _conv(fea, net.upconv1, _SLOPE, upsample=1)
    fea = _conv(fea, net.upconv2, _SLOPE, upsample=1)
    return _conv(_conv(fea, net.HR_conv, _SLOPE), net.conv_last, nchw_out=True)
