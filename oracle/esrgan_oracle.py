"""CPU oracle for the ESRGAN x4 generator hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, as plain functions over a ``state_dict`` on the CPU in fp32,
the arithmetic of the reference model definitions:

* dense block      -> /root/reference/esrgan/models/esrgan_model.py:18-24
* RRDB             -> /root/reference/esrgan/models/esrgan_model.py:35-39
* generator        -> /root/reference/esrgan/models/esrgan_model.py:55-62
* discriminator    -> /root/reference/esrgan/models/esrgan_model.py:64-87
* pixel-phase step -> /root/reference/esrgan/train.py:156-163 (L1 loss, backward)

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker.  The product path
(``super_resolution_amd``) never imports it and has no CPU fallback.

Pinning: ``tests/golden/*.npz`` were produced by importing the reference classes
themselves (``tests/golden/make_golden.py``, run in the build container where
``/root/reference`` exists) and ``tests/test_oracle.py`` checks every function
here against those vectors bit-for-bit.  The reference has no tests or golden
vectors of its own for this path.

``emulate_bf16`` is NOT part of the reference: it models the storage rounding
points of the bf16 HIP path (weights and every conv input rounded to bf16, fp32
accumulation, fp32 residual trunk; "hr_split" additionally models the convs
whose rounding residue the HIP path carries) so the bf16 kernels can be checked
against a like-for-like CPU computation besides the PSNR gate against fp32.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Iterable, List, Optional, Tuple

import torch
import torch.nn.functional as F

StateDict = Dict[str, torch.Tensor]

LRELU_SLOPE = 0.2   # esrgan_model.py:15, :53
RES_SCALE = 0.2     # esrgan_model.py:16, :33


# --------------------------------------------------------------------------
# parameter inventory (construction order == state_dict order of the reference)
# --------------------------------------------------------------------------
def generator_conv_specs(in_channels: int = 3, out_channels: int = 3, num_features: int = 64,
                         num_blocks: int = 23, growth_channels: int = 32
                         ) -> List[Tuple[str, int, int]]:
    """(name, cin, cout) of every 3x3 conv in constructor order (esrgan_model.py:42-53)."""
    nf, gc = num_features, growth_channels
    specs: List[Tuple[str, int, int]] = [("conv_first", in_channels, nf)]
    for b in range(num_blocks):
        for r in (1, 2, 3):
            for k in range(1, 6):
                cin = nf + gc * (k - 1)
                cout = gc if k < 5 else nf
                specs.append((f"body.{b}.rdb{r}.conv{k}", cin, cout))
    for name in ("trunk_conv", "upconv1", "upconv2", "HR_conv"):
        specs.append((name, nf, nf))
    specs.append(("conv_last", nf, out_channels))
    return specs


def make_conv_state(specs: Iterable[Tuple[str, int, int]], gen: torch.Generator) -> StateDict:
    """Draw (weight, bias) for each 3x3 conv exactly as ``torch.nn.Conv2d.reset_parameters`` does:
    kaiming-uniform weight with a=sqrt(5), then a uniform bias bounded by 1/sqrt(fan_in)."""
    sd: StateDict = OrderedDict()
    for name, cin, cout in specs:
        fan_in = cin * 9
        gain = math.sqrt(2.0 / (1 + math.sqrt(5) ** 2))   # leaky_relu gain with a=sqrt(5)
        w_bound = math.sqrt(3.0) * (gain / math.sqrt(fan_in))
        sd[name + ".weight"] = torch.empty(cout, cin, 3, 3).uniform_(-w_bound, w_bound, generator=gen)
        b_bound = 1.0 / math.sqrt(fan_in)
        sd[name + ".bias"] = torch.empty(cout).uniform_(-b_bound, b_bound, generator=gen)
    return sd


def make_generator_state(seed: int, **cfg) -> StateDict:
    """Seeded default-initialised generator weights: the numbers ``torch.manual_seed(seed)``
    followed by the reference constructor (esrgan_model.py:42-53) produces."""
    return make_conv_state(generator_conv_specs(**cfg), torch.Generator().manual_seed(seed))


def make_discriminator_state(seed: int, base: int = 64) -> StateDict:
    """Seeded default-initialised discriminator (esrgan_model.py:64-84).  BatchNorm draws nothing
    from the RNG (weight 1, bias 0, running stats 0/1)."""
    gen = torch.Generator().manual_seed(seed)
    sd: StateDict = OrderedDict()
    for ci, cin, cout, _stride, bi in discriminator_layout(base):
        sd.update(make_conv_state([(f"model.{ci}", cin, cout)], gen))
        if bi is not None:
            sd[f"model.{bi}.weight"] = torch.ones(cout)
            sd[f"model.{bi}.bias"] = torch.zeros(cout)
            sd[f"model.{bi}.running_mean"] = torch.zeros(cout)
            sd[f"model.{bi}.running_var"] = torch.ones(cout)
            sd[f"model.{bi}.num_batches_tracked"] = torch.tensor(0)
    return sd


def center_outputs(sd: StateDict, value: float = 0.5) -> StateDict:
    """Fixture recipe: default init drives ~96 % of outputs outside [0,1], which makes
    clamped comparisons meaningless; shifting ``conv_last.bias`` re-centres them."""
    out = OrderedDict((k, v.clone()) for k, v in sd.items())
    out["conv_last.bias"] += value
    return out


# --------------------------------------------------------------------------
# forward
# --------------------------------------------------------------------------
class _RoundBf16(torch.autograd.Function):
    """Round to bf16 in the forward pass, pass the gradient through unchanged.  (``t.to(bfloat16).to(float32)`` would
    round the GRADIENT to bf16 as well on its way back through the casts: an emulation whose backward is meant to be
    exact autograd must not, or a BatchNorm below amplifies the emulation's own rounding noise.)"""

    @staticmethod
    def forward(ctx, t):
        return t.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g


def _bf16r(t: torch.Tensor) -> torch.Tensor:
    return _RoundBf16.apply(t) if t.requires_grad else t.to(torch.bfloat16).to(torch.float32)


_HR_SIDE = ("upconv1", "upconv2", "HR_conv", "conv_last")


def _conv(sd: StateDict, name: str, x: torch.Tensor, emulate_bf16, stride: int = 1) -> torch.Tensor:
    """``emulate_bf16``: False = fp32 (the reference); True = every conv rounds weights and input to
    bf16; "hr_split" = as True, except that the HR-side convs carry their weights, and conv_first its weights and
    input, as a bf16 value plus the bf16 rounding of the residue (the residue-carrying modes of the HIP path); "hr_f16" = as
    "hr_split", except that the HR-side convs round weights and input to fp16 instead."""
    w, b = sd[name + ".weight"], sd[name + ".bias"]
    if emulate_bf16 == "hr_f16" and name in _HR_SIDE:
        return F.conv2d(x.to(torch.float16).float(), w.to(torch.float16).float(), b, stride=stride, padding=1)
    if emulate_bf16:
        split = emulate_bf16 in ("hr_split", "hr_f16")
        carried = split and (name in _HR_SIDE or name == "conv_first")
        if carried:
            # "exact" on the HIP path means a bf16 value plus the bf16 rounding of its residue (16 significant bits).
            # The difference from the fp32 value is 2^-17 relative, but this network amplifies it: a perturbed sum
            # that crosses a bf16 rounding boundary downstream moves that activation by a whole bf16 step.
            w_hi = _bf16r(w)
            w_lo = _bf16r(w - w_hi)
            if name == "conv_first":
                x_hi = _bf16r(x)
                x_lo = _bf16r(x - x_hi)
                # three of the four products; lo * lo (2^-18 of the result) is not computed
                return F.conv2d(x_hi, w_hi + w_lo, b, stride=stride, padding=1) + \
                    F.conv2d(x_lo, w_hi, None, stride=stride, padding=1)
            return F.conv2d(_bf16r(x), w_hi + w_lo, b, stride=stride, padding=1)
        w = _bf16r(w)
        x = _bf16r(x)
    return F.conv2d(x, w, b, stride=stride, padding=1)


def dense_block_forward(sd: StateDict, prefix: str, x: torch.Tensor, emulate_bf16=False) -> torch.Tensor:
    """esrgan_model.py:18-24 — four growth convs on the running concatenation, then conv5."""
    feats = [x]
    for k in range(1, 5):
        y = _conv(sd, f"{prefix}conv{k}", torch.cat(feats, 1), emulate_bf16)
        feats.append(F.leaky_relu(y, LRELU_SLOPE))
    x5 = _conv(sd, f"{prefix}conv5", torch.cat(feats, 1), emulate_bf16)
    return x + x5 * RES_SCALE


def rrdb_forward(sd: StateDict, prefix: str, x: torch.Tensor, emulate_bf16=False) -> torch.Tensor:
    """esrgan_model.py:35-39."""
    out = x
    for r in (1, 2, 3):
        out = dense_block_forward(sd, f"{prefix}rdb{r}.", out, emulate_bf16)
    return x + out * RES_SCALE


def count_blocks(sd: StateDict) -> int:
    n = 0
    while f"body.{n}.rdb1.conv1.weight" in sd:
        n += 1
    return n


def generator_forward(sd: StateDict, x: torch.Tensor, emulate_bf16=False) -> torch.Tensor:
    """esrgan_model.py:55-62.  ``x`` is [B,3,H,W] fp32; returns [B,3,4H,4W] fp32, unclamped."""
    fea = _conv(sd, "conv_first", x, emulate_bf16)
    body = fea
    for b in range(count_blocks(sd)):
        body = rrdb_forward(sd, f"body.{b}.", body, emulate_bf16)
    fea = fea + _conv(sd, "trunk_conv", body, emulate_bf16)
    up = F.interpolate(fea, scale_factor=2, mode="nearest")
    up = F.leaky_relu(_conv(sd, "upconv1", up, emulate_bf16), LRELU_SLOPE)
    up = F.interpolate(up, scale_factor=2, mode="nearest")
    up = F.leaky_relu(_conv(sd, "upconv2", up, emulate_bf16), LRELU_SLOPE)
    hr = F.leaky_relu(_conv(sd, "HR_conv", up, emulate_bf16), LRELU_SLOPE)
    return _conv(sd, "conv_last", hr, emulate_bf16)


def generator_forward_tiled(sd: StateDict, x: torch.Tensor, bands: int, halo: int) -> torch.Tensor:
    """Row-band inference with ``halo`` LR rows of overlap on interior edges.  Not in the
    reference (esrgan/test.py:71-72 runs the whole image); the checker for the sharded path."""
    B, C, H, W = x.shape
    edges = [round(i * H / bands) for i in range(bands + 1)]
    outs = []
    for i in range(bands):
        y0, y1 = edges[i], edges[i + 1]
        a, b = max(0, y0 - halo), min(H, y1 + halo)
        sr = generator_forward(sd, x[:, :, a:b, :])
        outs.append(sr[:, :, 4 * (y0 - a): 4 * (y0 - a) + 4 * (y1 - y0), :])
    return torch.cat(outs, 2)


# --------------------------------------------------------------------------
# backward (autograd over the functional restatement)
# --------------------------------------------------------------------------
def generator_grads(sd: StateDict, x: torch.Tensor, grad_out: torch.Tensor, input_grad: bool = False,
                    emulate_bf16=False) -> Tuple[torch.Tensor, StateDict, Optional[torch.Tensor]]:
    """d(sum(out * grad_out)) / d(params[, x]) — what ``out.backward(grad_out)`` leaves in ``.grad``.
    With ``emulate_bf16`` the forward rounds like the bf16 HIP path (the casts pass gradients through),
    so the LeakyReLU sign pattern matches the one the HIP backward sees."""
    leaves = OrderedDict((k, v.detach().clone().requires_grad_(True)) for k, v in sd.items())
    xin = x.detach().clone().requires_grad_(input_grad)
    out = generator_forward(leaves, xin, emulate_bf16)
    out.backward(grad_out)
    grads = OrderedDict((k, v.grad) for k, v in leaves.items())
    return out.detach(), grads, (xin.grad if input_grad else None)


def pixel_step_grads(sd: StateDict, lr: torch.Tensor, hr: torch.Tensor) -> Tuple[float, StateDict]:
    """Pixel pre-training step of esrgan/train.py:156-161: L1 (mean) between G(lr) and hr."""
    leaves = OrderedDict((k, v.detach().clone().requires_grad_(True)) for k, v in sd.items())
    loss = F.l1_loss(generator_forward(leaves, lr), hr)
    loss.backward()
    return float(loss.detach()), OrderedDict((k, v.grad) for k, v in leaves.items())


# --------------------------------------------------------------------------
# discriminator (esrgan_model.py:64-87): conv, [BN], LeakyReLU stack
# --------------------------------------------------------------------------
def discriminator_layout(base: int = 64) -> List[Tuple[int, int, int, int, Optional[int]]]:
    """(conv_index, cin, cout, stride, bn_index|None) following the Sequential numbering of
    esrgan_model.py:74-84: block 0 has no BN (conv 0, act 1); later blocks are conv, BN, act."""
    widths = [(3, base, 1), (base, base, 2), (base, 2 * base, 1), (2 * base, 2 * base, 2),
              (2 * base, 4 * base, 1), (4 * base, 4 * base, 2), (4 * base, 8 * base, 1),
              (8 * base, 8 * base, 2)]
    layout = []
    idx = 0
    for i, (cin, cout, s) in enumerate(widths):
        if i == 0:
            layout.append((idx, cin, cout, s, None))
            idx += 2
        else:
            layout.append((idx, cin, cout, s, idx + 1))
            idx += 3
    layout.append((idx, 8 * base, 1, 1, None))
    return layout


def discriminator_forward(sd: StateDict, img: torch.Tensor, training: bool, base: int = 64,
                          momentum: float = 0.1, eps: float = 1e-5, emulate_bf16: bool = False,
                          slope_masks=None) -> torch.Tensor:
    """``emulate_bf16`` (not reference behaviour) rounds every conv's weights and input to bf16, the rounding
    points of the HIP path (conv outputs feed BatchNorm in fp32 there).  ``slope_masks`` (not reference behaviour
    either; a test device): one boolean NCHW tensor per activation, True where the LeakyReLU passes its input
    unchanged; the activation then multiplies by 1 / 0.2 as the mask says instead of looking at its own input's
    sign.  With the masks of another forward pass of the same network this pins the piecewise-linear branch, so that
    two backward passes can be compared without the slope flips that tiny forward differences cause."""
    x = img
    layout = discriminator_layout(base)
    for j, (ci, _cin, _cout, stride, bi) in enumerate(layout):
        w = sd[f"model.{ci}.weight"]
        if emulate_bf16:
            w, x = _bf16r(w), _bf16r(x)
        x = F.conv2d(x, w, sd[f"model.{ci}.bias"], stride=stride, padding=1)
        if j == len(layout) - 1:
            break
        if bi is not None:
            if training and f"model.{bi}.num_batches_tracked" in sd:      # nn.BatchNorm2d counts training forwards
                sd[f"model.{bi}.num_batches_tracked"] += 1
            x = F.batch_norm(x, sd[f"model.{bi}.running_mean"], sd[f"model.{bi}.running_var"],
                             sd[f"model.{bi}.weight"], sd[f"model.{bi}.bias"],
                             training=training, momentum=momentum, eps=eps)
        if slope_masks is not None:
            x = x * torch.where(slope_masks[j], 1.0, LRELU_SLOPE)
        else:
            x = F.leaky_relu(x, LRELU_SLOPE)
    return x


# --------------------------------------------------------------------------
# VGG19 feature extractor (esrgan_model.py:90-103 -> torchvision vgg19.features[:35])
# --------------------------------------------------------------------------
# torchvision is a third-party dependency of the reference (requirements.txt, unpinned) and is not
# installed here, and its ImageNet weights cannot be fetched: the ARITHMETIC below restates torchvision's
# published vgg19 configuration 'E' (conv3x3+ReLU stacks with 2x2 max pools; features[:35] ends with
# conv5_4 before its ReLU); parity of the VGG numbers against torchvision itself is unpinned.
VGG19_CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M")


def vgg19_layout(feature_layer: int = 35) -> List[Tuple[str, int, int, int]]:
    """[(kind, index, cin, cout)] of features[:feature_layer]; kind in {"conv", "relu", "pool"}."""
    out, idx, cin = [], 0, 3
    for v in VGG19_CFG:
        if v == "M":
            out.append(("pool", idx, cin, cin))
            idx += 1
        else:
            out.append(("conv", idx, cin, v))
            out.append(("relu", idx + 1, v, v))
            idx += 2
            cin = v
    return out[:feature_layer]


def make_vgg_state(seed: int, feature_layer: int = 35) -> StateDict:
    """Seeded stand-in weights with the reference's keys (features.N.weight / .bias): He-normal over the
    fan-out as torchvision initialises an untrained vgg19, zero biases."""
    gen = torch.Generator().manual_seed(seed)
    sd: StateDict = OrderedDict()
    for kind, idx, cin, cout in vgg19_layout(feature_layer):
        if kind == "conv":
            std = math.sqrt(2.0 / (cout * 9))
            sd[f"features.{idx}.weight"] = torch.randn(cout, cin, 3, 3, generator=gen) * std
            sd[f"features.{idx}.bias"] = torch.zeros(cout)
    return sd


def vgg_features(sd: StateDict, x: torch.Tensor, feature_layer: int = 35, emulate_bf16: bool = False) -> torch.Tensor:
    """``emulate_bf16`` (not reference behaviour) rounds weights and every stored activation to bf16 where the HIP
    path does, so that ReLU signs and pooling winners are the ones that path sees."""
    layout = vgg19_layout(feature_layer)
    for j, (kind, idx, _cin, _cout) in enumerate(layout):
        if kind == "conv":
            w = sd[f"features.{idx}.weight"]
            if emulate_bf16:
                w, x = _bf16r(w), _bf16r(x)
            x = F.conv2d(x, w, sd[f"features.{idx}.bias"], padding=1)
            fused_relu = j + 1 < len(layout) and layout[j + 1][0] == "relu"
            if emulate_bf16 and not fused_relu:
                x = _bf16r(x)
        elif kind == "relu":
            x = F.relu(x)
            if emulate_bf16:
                x = _bf16r(x)
        else:
            x = F.max_pool2d(x, 2, 2)
    return x


# --------------------------------------------------------------------------
# adversarial training iteration (esrgan/train.py:196-223), without autocast / GradScaler
# --------------------------------------------------------------------------
def adversarial_step(sd_g: StateDict, sd_d: StateDict, sd_v: StateDict, lr_img: torch.Tensor, hr_img: torch.Tensor,
                     lr: float = 1e-4, betas: Tuple[float, float] = (0.9, 0.999)) -> dict:
    """Runs the discriminator update and then the generator update exactly in the order of the reference loop
    body, with fresh Adam optimizers.  Returns the losses, the gradients each backward left, and the parameters
    and BatchNorm buffers after both steps."""
    g = OrderedDict((k, v.detach().clone().requires_grad_(True)) for k, v in sd_g.items())
    d = OrderedDict((k, (v.detach().clone().requires_grad_(True) if v.dtype.is_floating_point and
                         not k.split(".")[-1].startswith("running") else v.detach().clone()))
                    for k, v in sd_d.items())
    d_params = [v for v in d.values() if v.requires_grad]
    opt_g = torch.optim.Adam(list(g.values()), lr=lr, betas=betas)
    opt_d = torch.optim.Adam(d_params, lr=lr, betas=betas)

    opt_d.zero_grad()
    fake = generator_forward(g, lr_img).detach()
    pred_real = discriminator_forward(d, hr_img, training=True)
    pred_fake = discriminator_forward(d, fake, training=True)
    loss_d = 0.5 * (F.mse_loss(pred_real, torch.ones_like(pred_real)) + F.mse_loss(pred_fake, torch.zeros_like(pred_fake)))
    loss_d.backward()
    grads_d = OrderedDict((k, v.grad.clone()) for k, v in d.items() if v.requires_grad)
    opt_d.step()

    opt_g.zero_grad()
    for v in d_params:
        v.grad = None
    fake = generator_forward(g, lr_img)
    pred_fake = discriminator_forward(d, fake, training=True)
    loss_gan = F.mse_loss(pred_fake, torch.ones_like(pred_fake))
    loss_pix = F.l1_loss(fake, hr_img)
    loss_per = F.l1_loss(vgg_features(sd_v, fake), vgg_features(sd_v, hr_img))
    loss_g = loss_pix + 0.01 * loss_per + 1e-3 * loss_gan
    loss_g.backward()
    grads_g = OrderedDict((k, v.grad.clone()) for k, v in g.items())
    opt_g.step()
    return {"loss_D": float(loss_d.detach()), "loss_G": float(loss_g.detach()), "loss_G_pix": float(loss_pix.detach()),
            "loss_G_per": float(loss_per.detach()), "loss_G_gan": float(loss_gan.detach()),
            "grads_D": grads_d, "grads_G": grads_g,
            "state_G": OrderedDict((k, v.detach()) for k, v in g.items()),
            "state_D": OrderedDict((k, v.detach()) for k, v in d.items())}


# --------------------------------------------------------------------------
# metrics used by the parity gates
# --------------------------------------------------------------------------
def _conv3(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, emulate_bf16: bool, stride: int = 1) -> torch.Tensor:
    """A 3x3 / pad 1 conv of the widened models; ``emulate_bf16`` rounds input and weight to bf16 where the HIP conv core
    does (the casts pass gradients through), so that a comparison of gradients is like for like."""
    if emulate_bf16:
        x, w = _bf16r(x), _bf16r(w)
    return F.conv2d(x, w, b, stride=stride, padding=1)


# ---- CycleGAN super-resolution generator (CycleGAN/CycleGAN_w_v1/CycleGAN.py:89-137; v2 is the same network) ----
def generator_sr_layout(in_channels: int = 3, out_channels: int = 3, num_residual: int = 6,
                        base_channels: int = 64) -> List[Tuple[str, int, int, int]]:
    """(state_dict prefix, cin, cout, kernel) of every conv in construction order (CycleGAN.py:105-127)."""
    b = base_channels
    convs = [("conv1.0", in_channels, b, 7)]
    for i in range(num_residual):
        convs += [(f"res_blocks.{i}.conv1", b, b, 3), (f"res_blocks.{i}.conv2", b, b, 3)]
    convs += [("up1.0", b, 4 * b, 3), ("up2.0", b, 4 * b, 3), ("conv_out", b, out_channels, 7)]
    return convs


def make_generator_sr_state(seed: int, **cfg) -> StateDict:
    """The weights ``torch.manual_seed(seed); GeneratorSR(**cfg)`` holds (default Conv2d initialisation, drawn
    in construction order)."""
    gen = torch.Generator().manual_seed(seed)
    sd: StateDict = OrderedDict()
    for name, cin, cout, k in generator_sr_layout(**cfg):
        fan_in = cin * k * k
        w_bound = math.sqrt(3.0) * (math.sqrt(2.0 / (1 + 5.0)) / math.sqrt(fan_in))
        sd[name + ".weight"] = torch.empty(cout, cin, k, k).uniform_(-w_bound, w_bound, generator=gen)
        sd[name + ".bias"] = torch.empty(cout).uniform_(-1.0 / math.sqrt(fan_in), 1.0 / math.sqrt(fan_in), generator=gen)
    return sd


def generator_sr_forward(sd: StateDict, x: torch.Tensor, emulate_bf16: bool = False) -> torch.Tensor:
    """GeneratorSR.forward (CycleGAN.py:129-137): 7x7 conv + ReLU, residual blocks ``x + conv2(relu(conv1(x)))``
    (:96-100), two (3x3 conv to 4x channels, PixelShuffle(2), ReLU) stages, 7x7 conv, tanh.  ``emulate_bf16``: the 3x3
    convs round their operands as the HIP path does (the residual trunk stays fp32 there too)."""
    h = F.relu(F.conv2d(x, sd["conv1.0.weight"], sd["conv1.0.bias"], padding=3))
    i = 0
    while f"res_blocks.{i}.conv1.weight" in sd:
        p = f"res_blocks.{i}."
        t = F.relu(_conv3(h, sd[p + "conv1.weight"], sd[p + "conv1.bias"], emulate_bf16))
        h = h + _conv3(t, sd[p + "conv2.weight"], sd[p + "conv2.bias"], emulate_bf16)
        i += 1
    for up in ("up1.0", "up2.0"):
        h = F.relu(F.pixel_shuffle(_conv3(h, sd[up + ".weight"], sd[up + ".bias"], emulate_bf16), 2))
    return torch.tanh(F.conv2d(h, sd["conv_out.weight"], sd["conv_out.bias"], padding=3))


# ---- CycleGAN HR -> LR generator (CycleGAN/CycleGAN_w_v1/CycleGAN.py:140-163; v2 is the same network) ----
def generator_down_layout(in_channels: int = 3, out_channels: int = 3,
                          base_channels: int = 64) -> List[Tuple[str, int, int, int]]:
    """(state_dict prefix, cin, cout, kernel) of every conv in construction order (CycleGAN.py:143-156)."""
    b = base_channels
    return [("down.0", in_channels, b, 7), ("down.2", b, 2 * b, 4), ("down.4", 2 * b, 4 * b, 4),
            ("res_blocks.0.conv1", 4 * b, 4 * b, 3), ("res_blocks.0.conv2", 4 * b, 4 * b, 3),
            ("res_blocks.1.conv1", 4 * b, 4 * b, 3), ("res_blocks.1.conv2", 4 * b, 4 * b, 3),
            ("out_conv", 4 * b, out_channels, 7)]


def make_generator_down_state(seed: int, **cfg) -> StateDict:
    """The weights ``torch.manual_seed(seed); GeneratorDown(**cfg)`` holds (default Conv2d initialisation)."""
    gen = torch.Generator().manual_seed(seed)
    sd: StateDict = OrderedDict()
    for name, cin, cout, k in generator_down_layout(**cfg):
        fan_in = cin * k * k
        w_bound = math.sqrt(3.0) * (math.sqrt(2.0 / (1 + 5.0)) / math.sqrt(fan_in))
        sd[name + ".weight"] = torch.empty(cout, cin, k, k).uniform_(-w_bound, w_bound, generator=gen)
        sd[name + ".bias"] = torch.empty(cout).uniform_(-1.0 / math.sqrt(fan_in), 1.0 / math.sqrt(fan_in), generator=gen)
    return sd


def generator_down_forward(sd: StateDict, x: torch.Tensor, emulate_bf16: bool = False) -> torch.Tensor:
    """GeneratorDown.forward (CycleGAN.py:158-163): 7x7 conv + ReLU, two (4x4 / stride 2 conv, ReLU) stages, two
    residual blocks, 7x7 conv, tanh.  ``emulate_bf16``: the residual blocks' 3x3 convs round their operands as the HIP
    path does."""
    h = F.relu(F.conv2d(x, sd["down.0.weight"], sd["down.0.bias"], padding=3))
    h = F.relu(F.conv2d(h, sd["down.2.weight"], sd["down.2.bias"], stride=2, padding=1))
    h = F.relu(F.conv2d(h, sd["down.4.weight"], sd["down.4.bias"], stride=2, padding=1))
    for i in range(2):
        p = f"res_blocks.{i}."
        t = F.relu(_conv3(h, sd[p + "conv1.weight"], sd[p + "conv1.bias"], emulate_bf16))
        h = h + _conv3(t, sd[p + "conv2.weight"], sd[p + "conv2.bias"], emulate_bf16)
    return torch.tanh(F.conv2d(h, sd["out_conv.weight"], sd["out_conv.bias"], padding=3))


# ---- conditional VAE ("condtional VAE/model.py":53-215) ----
def _bn(sd: StateDict, name: str, x: torch.Tensor, training: bool) -> torch.Tensor:
    """nn.BatchNorm2d as the reference applies it (default eps / momentum); in training mode the running
    buffers of ``sd`` are updated in place, as the module does."""
    if training:
        with torch.no_grad():
            mean = x.mean((0, 2, 3))
            var_u = x.var((0, 2, 3), unbiased=True)
            sd[name + ".running_mean"].mul_(0.9).add_(0.1 * mean)
            sd[name + ".running_var"].mul_(0.9).add_(0.1 * var_u)
            sd[name + ".num_batches_tracked"] += 1
    return F.batch_norm(x, None if training else sd[name + ".running_mean"], None if training else sd[name + ".running_var"],
                        sd[name + ".weight"], sd[name + ".bias"], training, 0.1, 1e-5)


def make_cvae_state(seed: int, latent_dim: int = 256) -> StateDict:
    """The state ``torch.manual_seed(seed); CVAE(latent_dim)`` holds: default initialisation of every Conv2d, Linear
    and ConvTranspose2d in construction order (model.py:58-147); BatchNorm draws nothing."""
    gen = torch.Generator().manual_seed(seed)
    sd: StateDict = OrderedDict()

    def draw(name, shape, fan_in):
        bound = math.sqrt(3.0) * (math.sqrt(2.0 / 6.0) / math.sqrt(fan_in))
        sd[name + ".weight"] = torch.empty(*shape).uniform_(-bound, bound, generator=gen)
        b = 1.0 / math.sqrt(fan_in)
        n_bias = shape[1] if name.startswith("dec_conv") else shape[0]
        sd[name + ".bias"] = torch.empty(n_bias).uniform_(-b, b, generator=gen)

    def bn(name, c):
        sd[name + ".weight"], sd[name + ".bias"] = torch.ones(c), torch.zeros(c)
        sd[name + ".running_mean"], sd[name + ".running_var"] = torch.zeros(c), torch.ones(c)
        sd[name + ".num_batches_tracked"] = torch.tensor(0)

    draw("enc_conv1.0", (64, 3, 3, 3), 27)
    for k, (ci, co) in zip((2, 3, 4), ((64, 128), (128, 256), (256, 512))):
        draw(f"enc_conv{k}.0", (co, ci, 3, 3), ci * 9)
        bn(f"enc_conv{k}.1", co)
    draw("fc_mu", (latent_dim, 8192), 8192)
    draw("fc_logvar", (latent_dim, 8192), 8192)
    draw("decoder_input", (8192, latent_dim), latent_dim)
    draw("condition_processor.0", (32, 3, 3, 3), 27)
    draw("condition_processor.2", (64, 32, 3, 3), 32 * 9)
    # ConvTranspose2d weights are [in, out, k, k]; its fan_in is taken from dimension 1 (out * k * k)
    for k, (ci, co) in zip((1, 2, 3), ((512, 256), (512, 128), (256, 64))):
        draw(f"dec_conv{k}.0", (ci, co, 4, 4), co * 16)
        bn(f"dec_conv{k}.1", co)
    draw("upsample1.0", (256, 128, 3, 3), 128 * 9)
    bn("upsample1.2", 64)
    draw("upsample2.0", (256, 64, 3, 3), 64 * 9)
    bn("upsample2.2", 64)
    draw("final_conv.0", (3, 64, 3, 3), 64 * 9)
    draw("lr_upsampler.1", (3, 3, 3, 3), 27)
    return sd


def cvae_encode(sd: StateDict, x: torch.Tensor, training: bool = False, emulate_bf16: bool = False):
    """CVAE.encode (model.py:150-163): (mu, logvar, (x1, x2, x3, x4)).  ``emulate_bf16``: the 3x3 convs round their
    operands as the HIP path does."""
    x1 = F.leaky_relu(_conv3(x, sd["enc_conv1.0.weight"], sd["enc_conv1.0.bias"], emulate_bf16), 0.2)
    feats, h = [x1], x1
    for k in (2, 3, 4):
        h = _conv3(h, sd[f"enc_conv{k}.0.weight"], sd[f"enc_conv{k}.0.bias"], emulate_bf16, stride=2)
        h = F.leaky_relu(_bn(sd, f"enc_conv{k}.1", h, training), 0.2)
        feats.append(h)
    flat = h.flatten(1)
    mu = F.linear(flat, sd["fc_mu.weight"], sd["fc_mu.bias"])
    logvar = F.linear(flat, sd["fc_logvar.weight"], sd["fc_logvar.bias"])
    return mu, logvar, tuple(feats)


def cvae_decode(sd: StateDict, z: torch.Tensor, condition: torch.Tensor, skips, training: bool = False,
                emulate_bf16: bool = False) -> torch.Tensor:
    """CVAE.decode (model.py:170-204).  ``condition_processor`` (:99-104) is evaluated by the reference and its
    result never used; it is left out here."""
    skip1, skip2, skip3, _skip4 = skips
    h = F.linear(z, sd["decoder_input.weight"], sd["decoder_input.bias"]).view(-1, 512, 4, 4)
    for k, skip in ((1, skip3), (2, skip2), (3, skip1)):
        h = F.conv_transpose2d(h, sd[f"dec_conv{k}.0.weight"], sd[f"dec_conv{k}.0.bias"], stride=2, padding=1)
        h = F.leaky_relu(_bn(sd, f"dec_conv{k}.1", h, training), 0.2)
        h = torch.cat([h, skip], 1)
    for k in (1, 2):
        h = F.pixel_shuffle(_conv3(h, sd[f"upsample{k}.0.weight"], sd[f"upsample{k}.0.bias"], emulate_bf16), 2)
        h = F.leaky_relu(_bn(sd, f"upsample{k}.2", h, training), 0.2)
    h = torch.tanh(_conv3(h, sd["final_conv.0.weight"], sd["final_conv.0.bias"], emulate_bf16))
    up = F.interpolate(condition, scale_factor=4, mode="bicubic", align_corners=False)
    return h + _conv3(up, sd["lr_upsampler.1.weight"], sd["lr_upsampler.1.bias"], emulate_bf16)


def psnr(a: torch.Tensor, b: torch.Tensor, peak: float = 1.0) -> float:
    mse = torch.mean((a.double() - b.double()) ** 2).item()
    return float("inf") if mse == 0 else 10.0 * math.log10(peak * peak / mse)


def inside_unit_fraction(t: torch.Tensor) -> float:
    return ((t > 0) & (t < 1)).double().mean().item()
