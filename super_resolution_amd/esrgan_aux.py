"""Discriminator and VGG19 feature extractor of the reference on the HIP conv core, and its two training steps.

Drop-ins for /root/reference/esrgan/models/esrgan_model.py:64-103: same class names, constructor
arguments and ``state_dict`` keys (the parameters live in the same ``nn.Sequential`` containers), so
checkpoints interchange.  Every 3x3 convolution runs through ``conv_core.conv3x3`` (the generator's MFMA
kernels, forward / data gradient / weight gradient; stride 2 computed at the output positions); BatchNorm +
LeakyReLU and max pooling are HIP kernels on the blocked layout as well (``conv_core.conv_bn_act``,
``conv_core.max_pool2x2``), forward and backward, and the L1 / MSE terms of the two training iterations are one
reduction kernel each (``conv_core.l1_loss`` / ``mse_loss``).  No CPU path.

``pixel_step`` / ``adversarial_step`` restate the two loop bodies of /root/reference/esrgan/train.py
(:157-164 and :196-223) as functions of the same objects the script holds.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor
from torch.autograd.function import once_differentiable

from . import conv_core as cc
from . import ops
from ._lib import DTYPE_BF16


def _need_cuda(x: Tensor, who: str) -> None:
    if not x.is_cuda:
        raise RuntimeError(f"{who} runs only on a ROCm device (got a {x.device.type} tensor)")
    if x.dim() != 4:
        raise RuntimeError(f"{who}: expected an NCHW tensor, got {tuple(x.shape)}")
    if not x.is_floating_point():
        # nn.Conv2d refuses it the same way
        raise RuntimeError(f"{who}: Input type ({x.dtype}) and weight type (torch.float32) should be the same")
    if x.shape[0] and (x.shape[2] == 0 or x.shape[3] == 0):
        raise RuntimeError(f"{who}: Kernel size (3 x 3) can't be greater than the padded input size "
                           f"({x.shape[2] + 2} x {x.shape[3] + 2})")


class Discriminator(nn.Module):
    """esrgan_model.py:64-87 — nine 3x3 convs (strides 1,2,1,2,1,2,1,2,1), BatchNorm on all but the first and the
    last, LeakyReLU(0.2); [B,3,H,W] -> [B,1,H/16,W/16]."""

    def __init__(self, in_channels: int = 3, base_channels: int = 64):
        super().__init__()
        if base_channels % 16:
            raise ValueError("base_channels must be a multiple of 16 (the normalised layers run on 16-channel groups)")

        def conv_block(in_f, out_f, stride=1, bn=True):
            layers = [nn.Conv2d(in_f, out_f, 3, stride, 1)]
            if bn:
                layers.append(nn.BatchNorm2d(out_f))
            layers.append(nn.LeakyReLU(0.2, inplace=True))
            return layers

        b = base_channels
        self.model = nn.Sequential(
            *conv_block(in_channels, b, bn=False),
            *conv_block(b, b, stride=2),
            *conv_block(b, b * 2),
            *conv_block(b * 2, b * 2, stride=2),
            *conv_block(b * 2, b * 4),
            *conv_block(b * 4, b * 4, stride=2),
            *conv_block(b * 4, b * 8),
            *conv_block(b * 8, b * 8, stride=2),
            nn.Conv2d(b * 8, 1, 3, 1, 1),
        )

    def _layers(self):
        """(conv, bn or None, LeakyReLU slope or None) per conv of ``self.model``, in order; the last conv has neither."""
        mods, out, i = list(self.model), [], 0
        while i < len(mods):
            conv = mods[i]
            i += 1
            bn = slope = None
            if i < len(mods) and isinstance(mods[i], nn.BatchNorm2d):
                bn = mods[i]
                i += 1
            if i < len(mods) and isinstance(mods[i], nn.LeakyReLU):
                slope = float(mods[i].negative_slope)
                i += 1
            if not isinstance(conv, nn.Conv2d) or (bn is not None and slope is None):
                raise RuntimeError("Discriminator: unexpected layer order in self.model")
            out.append((conv, bn, slope))
        if not out or out[-1][1] is not None or out[-1][2] is not None or any(sl is None for _c, _b, sl in out[:-1]):
            raise RuntimeError("Discriminator: expected conv blocks with an activation and one final plain conv")
        return out

    def _check_batch(self, n: int, h: int, w: int) -> None:
        """nn.BatchNorm2d's own refusal of a training batch with one value per channel (the deepest normalised layer
        sees the fewest)."""
        for _k in range(4):
            h, w = (h + 1) // 2, (w + 1) // 2
        if self.training and n * h * w == 1 and any(isinstance(m, nn.BatchNorm2d) and m.training for m in self.model):
            raise ValueError(f"Expected more than 1 value per channel when training, got input size "
                             f"torch.Size([{n}, {self.model[-1].in_channels}, {h}, {w}])")

    @staticmethod
    def _empty_result(img: Tensor) -> Tensor:
        h, w = int(img.shape[2]), int(img.shape[3])
        for _k in range(4):
            h, w = (h + 1) // 2, (w + 1) // 2
        return img.new_zeros((0, 1, h, w), dtype=torch.float32)

    def forward(self, img: Tensor) -> Tensor:
        _need_cuda(img, "Discriminator")
        if img.shape[0] == 0:
            return self._empty_result(img)
        self._check_batch(int(img.shape[0]), int(img.shape[2]), int(img.shape[3]))
        params = [p for p in self.parameters()]
        cc.prepare_packs([m for m in self.model if isinstance(m, nn.Conv2d)], torch.is_grad_enabled())
        if torch.is_grad_enabled() and (img.requires_grad or any(p.requires_grad for p in params)):
            # recorded: the whole network is one autograd node whose backward keeps the gradients between the layers
            # in fp32 wherever a BatchNorm's backward consumes them
            return _DiscriminatorFunction.apply(img, self, 1, *params)
        return self._forward_layers(img)

    def forward_pair(self, real: Tensor, fake: Tensor) -> Tuple[Tensor, Tensor]:
        """``(self(real), self(fake))`` — the two calls of the discriminator update (train.py:200-201) — as ONE pass over the
        concatenated batch: every conv (and its data / weight gradient) runs once on both halves, every BatchNorm takes
        its batch statistics per half and updates the running buffers with the real half first, then the fake half,
        exactly as two calls do.  Same predictions and gradients, half the conv launches, no second accumulation into
        ``.grad``."""
        _need_cuda(real, "Discriminator")
        if real.shape != fake.shape or real.device != fake.device:
            raise RuntimeError("Discriminator.forward_pair: two batches of one shape on one device")
        if real.shape[0] == 0:
            return self._empty_result(real), self._empty_result(fake)
        self._check_batch(int(real.shape[0]), int(real.shape[2]), int(real.shape[3]))
        params = [p for p in self.parameters()]
        cc.prepare_packs([m for m in self.model if isinstance(m, nn.Conv2d)], torch.is_grad_enabled())
        n = int(real.shape[0])
        both = torch.cat([real, fake], dim=0)
        if torch.is_grad_enabled() and (both.requires_grad or any(p.requires_grad for p in params)):
            pred = _DiscriminatorFunction.apply(both, self, 2, *params)
        else:
            pred = torch.cat([self._forward_layers(real), self._forward_layers(fake)], dim=0)
        return pred[:n], pred[n:]

    def _forward_layers(self, img: Tensor) -> Tensor:
        mods = list(self.model)
        x = cc.to_blocked(img)
        i = 0
        while i < len(mods):
            conv = mods[i]
            i += 1
            if i == len(mods):                                           # final 1-channel conv, no activation
                return cc.conv3x3(x, conv.weight, conv.bias, 1.0, nchw_out=True)
            if isinstance(mods[i], nn.BatchNorm2d):
                bn, act = mods[i], mods[i + 1]
                i += 2
                # conv (stride 1 or 2, fp32 out) -> batch statistics / running buffers as nn.BatchNorm2d -> LeakyReLU,
                # blocked bf16 out: three HIP stages, no layout shuffle
                x = cc.conv_bn_act(x, conv, bn, act.negative_slope)
            else:
                act = mods[i]
                i += 1
                x = cc.conv3x3(x, conv.weight, conv.bias, act.negative_slope, stride=conv.stride[0])   # LeakyReLU fused
        raise RuntimeError("unreachable")


class _DiscriminatorFunction(torch.autograd.Function):
    """The discriminator (esrgan_model.py:64-87) as ONE autograd node: the forward runs the nine convs (stride 2 at the
    output positions, fp32 into every BatchNorm) and the BatchNorm + LeakyReLU kernels on the blocked layout and keeps
    what the backward needs; the backward (autograd of esrgan/train.py:205, :220) walks the layers in reverse and hands
    the data gradient from layer to layer as an fp32 tensor wherever a BatchNorm's backward consumes it.  BatchNorm's
    backward subtracts the gradient's mean over the batch and the map, and the adversarial losses' gradients are
    nearly constant there: on bf16-rounded gradients the small remainder is mostly rounding noise (6-7 % error on every
    parameter gradient with one autograd node per layer), on fp32 ones it is not."""

    @staticmethod
    def forward(ctx, img: Tensor, module: "Discriminator", groups: int, *params: Tensor) -> Tensor:
        # groups: the batch is that many equal parts, each a batch of its own to every BatchNorm (statistics, running
        # buffers updated part by part in order); convs and their gradients run over the whole batch
        layers = module._layers()
        if groups < 1 or img.shape[0] % groups:
            raise RuntimeError("Discriminator: the batch does not split into the requested groups")
        part = int(img.shape[0]) // groups
        x = ops.to_blocked_now(img.detach().float().contiguous(), cc.units_for(img.shape[1]), DTYPE_BF16)
        saved = []
        out = None
        for k, (conv, bn, slope) in enumerate(layers):
            w, b, stride = conv.weight, conv.bias, int(conv.stride[0])
            if k == len(layers) - 1:
                out = cc._conv_forward(x, w, b, 1.0, "nchw", stride)
                saved.append((x,))
            elif bn is None:
                y = cc._conv_forward(x, w, b, slope, "blocked", stride)
                saved.append((x, y))
                x = y
            else:
                cout = int(w.shape[0])
                if cout % 16:
                    raise RuntimeError("Discriminator: a normalised layer needs a multiple of 16 channels")
                if bn.momentum is None or not bn.track_running_stats or not bn.affine:
                    raise RuntimeError("Discriminator: BatchNorm2d with affine parameters, running statistics and a fixed "
                                       "momentum")
                y32 = cc._conv_forward(x, w, b, 1.0, "f32", stride)
                N, H, _q, W, _c = y32.shape
                mean = torch.empty((groups, cout), dtype=torch.float32, device=x.device)
                rstd = torch.empty((groups, cout), dtype=torch.float32, device=x.device)
                act = torch.empty((N, H, cout // 8, W, 8), dtype=torch.bfloat16, device=x.device)
                training = bool(bn.training)
                # (one set of launches for all groups: statistics per group, the running buffers updated group by group)
                ops.bn_lrelu_fwd_now(y32, bn.weight.detach(), bn.bias.detach(), float(bn.eps), float(slope), training,
                                     float(bn.momentum), bn.running_mean, bn.running_var, bn.num_batches_tracked, mean,
                                     rstd, act, 0, groups)
                saved.append((x, y32, mean, rstd, training))
                x = act
        ctx.layers, ctx.saved, ctx.in_channels, ctx.groups, ctx.part = layers, saved, int(img.shape[1]), groups, part
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dy: Tensor):
        layers, saved = ctx.layers, ctx.saved
        need = ctx.needs_input_grad          # (img, module, groups, *params)
        groups, part = ctx.groups, ctx.part
        grads = []                           # per layer, in reverse: its parameters' gradients in parameter order
        g = None
        p_at = 3 + sum(2 if bn is None else 4 for _c, bn, _s in layers)       # one past the last parameter
        for k in range(len(layers) - 1, -1, -1):
            conv, bn, slope = layers[k]
            w, b, stride = conv.weight, conv.bias, int(conv.stride[0])
            n_par = 2 if bn is None else 4
            p_at -= n_par
            need_dw = bool(need[p_at] or need[p_at + 1])
            need_dx = k > 0 or bool(need[0])
            dx_f32 = k > 0 and layers[k - 1][1] is not None        # the layer below ends in a BatchNorm
            x = saved[k][0]
            dgamma = dbeta = None
            dy_split = False
            if k == len(layers) - 1:
                # the loss's gradient on the one-channel prediction map as hi + lo bf16 pairs: it is nearly constant
                # over the map, the BatchNorm below removes the constant, and what is left must not be rounding noise
                dy_split = stride == 1 and cc.split_cotangent_fits(int(w.shape[0]))
                dyb = ops.to_blocked_now(dy.float().contiguous(), cc.units_for(int(w.shape[0])), DTYPE_BF16,
                                         1 if dy_split else 0)
            elif bn is None:
                y = saved[k][1]
                gb = g if g.dtype == torch.bfloat16 else cc._f32_layout_to_blocked(g)
                dyb = (torch.ops.aten.threshold_backward(gb, y, 0.0) if slope == 0.0
                       else torch.ops.aten.leaky_relu_backward(gb, y, slope, False))
            else:
                _x, y32, mean, rstd, training = saved[k]
                cout = int(w.shape[0])
                N, H, _q, W, _c = y32.shape
                dyb = torch.empty((N, H, cout // 8, W, 8), dtype=torch.bfloat16, device=y32.device)
                dgb = torch.empty((2, groups, cout), dtype=torch.float32, device=y32.device)
                dgamma, dbeta = dgb[0], dgb[1]
                g = g.contiguous()
                ops.bn_lrelu_bwd_now(y32, g, bn.weight.detach(), bn.bias.detach(), mean, rstd, float(slope), training,
                                     dyb, dgamma, dbeta, 0, 0, groups)
                if groups == 1:
                    dgamma, dbeta = dgamma[0], dbeta[0]
                else:                                  # the groups' contributions add up, as two calls would accumulate
                    dgamma, dbeta = dgb.sum(1).unbind(0)
            dx, dw, db = cc._conv_backward(x, w, b, dyb, stride, need_dx, need_dw, dx_f32=dx_f32, dy_split=dy_split)
            grads.append([dw, db] if bn is None else [dw, db, dgamma, dbeta])
            g = dx
        d_img = ops.from_blocked_now(g.contiguous(), 0, ctx.in_channels) if need[0] else None
        flat = [t for layer in reversed(grads) for t in layer]
        return (d_img, None, None, *flat)


_VGG19 = (64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M")


def _vgg19_feature_layers() -> list:
    layers, cin = [], 3
    for v in _VGG19:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    return layers


class VGGFeatureExtractor(nn.Module):
    """esrgan_model.py:90-103 — ``vgg19.features[:feature_layer]`` (default 35: up to conv5_4, before its ReLU),
    frozen.  ``use_bn`` changes nothing, as in the reference (:94-97 build the same plain vgg19 either way).

    The ImageNet weights come from torchvision as in the reference (:93); where torchvision is not installed
    pass ``pretrained=False`` for the architecture with default-initialised weights and load a ``state_dict``
    with the reference's keys (``features.N.weight``)."""

    def __init__(self, feature_layer: int = 35, use_bn: bool = False, *, pretrained: bool = True):
        super().__init__()
        layers = _vgg19_feature_layers()
        if pretrained:
            try:
                from torchvision import models
            except ImportError as exc:
                raise ImportError("VGGFeatureExtractor(pretrained=True) takes the ImageNet weights from torchvision, "
                                  "which is not installed; use pretrained=False and load_state_dict") from exc
            src = models.vgg19(pretrained=True).features
            for dst, s in zip(layers, src):
                if isinstance(dst, nn.Conv2d):
                    dst.load_state_dict(s.state_dict())
        self.features = nn.Sequential(*layers[:feature_layer])
        for p in self.features.parameters():
            p.requires_grad = False

    def forward(self, x: Tensor) -> Tensor:
        _need_cuda(x, "VGGFeatureExtractor")
        if x.shape[0] == 0:
            return self._empty_result(x)
        mods = list(self.features)
        h = cc.to_blocked(x)
        channels = x.shape[1]
        i = 0
        while i < len(mods):
            m = mods[i]
            i += 1
            if isinstance(m, nn.Conv2d):
                relu = i < len(mods) and isinstance(mods[i], nn.ReLU)
                h = cc.conv3x3(h, m.weight, m.bias, 0.0 if relu else 1.0)    # ReLU = LeakyReLU with slope 0, fused
                channels = m.out_channels
                if relu:
                    i += 1
            elif isinstance(m, nn.MaxPool2d):
                h = cc.max_pool2x2(h)
            elif isinstance(m, nn.ReLU):
                h = torch.relu(h)
            else:
                raise RuntimeError(f"unexpected layer {type(m).__name__} in the VGG feature stack")
        return cc.from_blocked(h, channels)


    def _empty_result(self, x: Tensor) -> Tensor:
        """The features of a batch of no images: ``[0, C, h, w]`` with the stack's output geometry."""
        c, h, w = int(x.shape[1]), int(x.shape[2]), int(x.shape[3])
        for m in self.features:
            if isinstance(m, nn.Conv2d):
                c = m.out_channels
            elif isinstance(m, nn.MaxPool2d):
                h, w = h // 2, w // 2
        return x.new_zeros((0, c, h, w), dtype=torch.float32)

    def _plan(self) -> list:
        """The feature stack as (kind, module, relu) steps: "conv" with its fused ReLU flag, "pool"."""
        mods = list(self.features)
        plan, i = [], 0
        while i < len(mods):
            m = mods[i]
            i += 1
            if isinstance(m, nn.Conv2d):
                relu = i < len(mods) and isinstance(mods[i], nn.ReLU)
                plan.append(("conv", m, relu))
                if relu:
                    i += 1
            elif isinstance(m, nn.MaxPool2d):
                plan.append(("pool", m, False))
            else:
                raise RuntimeError(f"unexpected layer {type(m).__name__} in the VGG feature stack")
        return plan

    def forward_pair(self, sr: Tensor, hr: Tensor) -> Tuple[Tensor, Tensor]:
        """``(self(sr), self(hr).detach())`` — the two calls of the perceptual term (train.py:214-216) — as ONE forward
        over the concatenated batch: the extractor has no batch statistics, so every image's features are what the
        separate calls give, the launches are half as many and twice as full (the deep layers see 8 x 8 pixel maps);
        the backward runs the data gradients of the ``sr`` half only."""
        _need_cuda(sr, "VGGFeatureExtractor")
        if sr.shape != hr.shape or sr.device != hr.device:
            raise RuntimeError("VGGFeatureExtractor.forward_pair: two batches of one shape on one device")
        if sr.shape[0] == 0:
            return self._empty_result(sr) + sr.float().sum() * 0, self._empty_result(hr)
        return _VGGPairFunction.apply(sr, hr, self)


class _VGGPairFunction(torch.autograd.Function):
    """The frozen VGG19 feature stack (esrgan_model.py:90-103) on ``cat(sr, hr)`` as one autograd node: the forward keeps,
    for the ``sr`` half, each conv's input (and output where a ReLU is fused) and each pool's input; the backward walks
    the layers in reverse for that half only (ReLU gate, data gradient of the conv, first-maximum pooling gradient).
    The ``hr`` half of the result is marked non-differentiable (train.py:216 detaches it)."""

    @staticmethod
    def forward(ctx, sr: Tensor, hr: Tensor, module: "VGGFeatureExtractor"):
        n = int(sr.shape[0])
        x = torch.cat([sr.detach().float(), hr.detach().float()], dim=0).contiguous()
        h = ops.to_blocked_now(x, cc.units_for(x.shape[1]), DTYPE_BF16)
        saved, channels = [], int(x.shape[1])
        for kind, m, relu in module._plan():
            if kind == "conv":
                y = cc._conv_forward(h, m.weight, m.bias, 0.0 if relu else 1.0, "blocked", 1)
                saved.append(("conv", m, h[:n], y[:n] if relu else None))
                h, channels = y, m.out_channels
            else:
                saved.append(("pool", m, h[:n], None))
                h = ops.maxpool2x2_fwd_now(h)
        out = ops.from_blocked_now(h, 0, channels)
        ctx.saved, ctx.in_channels, ctx.units = saved, int(x.shape[1]), int(h.shape[2])
        f_sr, f_hr = out[:n], out[n:]
        ctx.mark_non_differentiable(f_hr)
        return f_sr, f_hr

    @staticmethod
    @once_differentiable
    def backward(ctx, g_sr: Tensor, _g_hr):
        if not ctx.needs_input_grad[0]:
            return None, None, None
        g = ops.to_blocked_now(g_sr.float().contiguous(), ctx.units, DTYPE_BF16)
        gated = False          # g already carries the derivative of the ReLU fused into the layer being visited
        for k in range(len(ctx.saved) - 1, -1, -1):
            kind, m, x, y = ctx.saved[k]
            if kind == "pool":
                g = ops.maxpool2x2_bwd_now(x.contiguous(), g.contiguous())
                gated = False
                continue
            if y is not None and not gated:
                g = torch.ops.aten.threshold_backward(g.contiguous(), y, 0.0)
            # where the layer below is a conv with a fused ReLU, this conv's input IS that ReLU's output: its
            # derivative gates the data gradient in the conv's epilogue instead of in a pass of its own
            below = ctx.saved[k - 1] if k > 0 else None
            fuse = below is not None and below[0] == "conv" and below[3] is not None
            g, _dw, _db = cc._conv_backward(x.contiguous(), m.weight, m.bias, g.contiguous(), 1, True, False,
                                            gate=x.contiguous() if fuse else None, gate_slope=0.0)
            gated = fuse
        return ops.from_blocked_now(g.contiguous(), 0, ctx.in_channels), None, None


# ------------------------------------------------------------------------------------------------
def _backward_and_step(loss: Tensor, optimizer, scaler) -> None:
    if scaler is None:
        loss.backward()
        optimizer.step()
    else:                       # train.py:161-163 / :205-207 / :220-222
        scaler.scale(loss).backward()
        scaler.step(optimizer)
        scaler.update()


def pixel_step(netG: nn.Module, optim_G, lr: Tensor, hr: Tensor, scaler=None) -> Dict[str, Tensor]:
    """One iteration of the pixel-loss pre-training loop (train.py:157-163): L1(netG(lr), hr), backward, Adam."""
    optim_G.zero_grad()
    sr = netG(lr)
    loss_pix = cc.l1_loss(sr, hr)
    _backward_and_step(loss_pix, optim_G, scaler)
    return {"loss_pix": loss_pix.detach()}


def _perceptual_features(vgg: nn.Module, fake: Tensor, hr: Tensor) -> Tuple[Tensor, Tensor]:
    """vgg(fake), vgg(hr) (train.py:214-216); one batched forward where the extractor offers it."""
    pair = getattr(vgg, "forward_pair", None)
    if pair is not None and fake.shape == hr.shape:
        return pair(fake, hr)
    return vgg(fake), vgg(hr)


def adversarial_step(netG: nn.Module, netD: nn.Module, vgg: nn.Module, optim_G, optim_D, lr: Tensor, hr: Tensor,
                     scaler_G=None, scaler_D=None) -> Dict[str, Tensor]:
    """One iteration of the adversarial loop (train.py:196-223): the discriminator update on (hr, detached fake),
    then the generator update with loss_G = L1 + 0.01 * L1(vgg) + 1e-3 * MSE(D(fake), 1).

    The reference runs the generator twice on the same batch (:198 for the discriminator's detached fake, :211 for
    its own loss); its weights do not change in between and it has no stochastic layer, so both calls return the same
    tensor.  Here it runs once, recorded, and the discriminator update takes the detached result: one generator
    forward less per iteration, and both updates see a fake image computed with the same arithmetic."""
    optim_D.zero_grad()
    optim_G.zero_grad()
    fake = netG(lr)
    pair = getattr(netD, "forward_pair", None)
    if pair is not None and hr.shape == fake.shape:
        pred_real, pred_fake = pair(hr, fake.detach())       # one pass over both batches, BatchNorm per batch
    else:
        pred_real = netD(hr)
        pred_fake = netD(fake.detach())
    loss_D = 0.5 * (cc.mse_loss(pred_real, 1.0) + cc.mse_loss(pred_fake, 0.0))
    begin = getattr(optim_D, "begin_step", None) if scaler_D is None else None
    if begin is None:
        _backward_and_step(loss_D, optim_D, scaler_D)
        feat_sr, feat_hr = _perceptual_features(vgg, fake, hr)
    else:
        # data parallel: the two VGG19 forwards need neither D's gradients nor its new weights, so they are enqueued
        # while the all-reduce of D's gradients is in flight
        loss_D.backward()
        begin()
        feat_sr, feat_hr = _perceptual_features(vgg, fake, hr)
        optim_D.step()

    # The generator's loss needs the discriminator's input gradient only.  The reference's loss_G.backward() (:220) also
    # accumulates gradients into netD's parameters, which nothing reads: the next iteration starts with
    # optim_D.zero_grad() (:197).  They are not computed here (a third of the discriminator's weight-gradient work).
    d_params = [p for p in netD.parameters() if p.requires_grad]
    for p in d_params:
        p.requires_grad_(False)
    try:
        pred_fake = netD(fake)
    finally:
        for p in d_params:
            p.requires_grad_(True)
    loss_G_gan = cc.mse_loss(pred_fake, 1.0)
    loss_G_pix = cc.l1_loss(fake, hr)
    loss_G_per = cc.l1_loss(feat_sr, feat_hr)
    loss_G = loss_G_pix + 0.01 * loss_G_per + 1e-3 * loss_G_gan
    _backward_and_step(loss_G, optim_G, scaler_G)
    return {"loss_D": loss_D.detach(), "loss_G": loss_G.detach(), "loss_G_pix": loss_G_pix.detach(),
            "loss_G_per": loss_G_per.detach(), "loss_G_gan": loss_G_gan.detach()}
