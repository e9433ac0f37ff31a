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
def _bf16r(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).to(torch.float32)


_HR_SIDE = ("upconv1", "upconv2", "HR_conv", "conv_last")


def _conv(sd: StateDict, name: str, x: torch.Tensor, emulate_bf16, stride: int = 1) -> torch.Tensor:
    """``emulate_bf16``: False = fp32 (the reference); True = every conv rounds weights and input to
    bf16; "hr_split" = as True, except that the HR-side convs keep their weights exact and
    conv_first keeps weights and input exact (the residue-carrying modes of the HIP path)."""
    w, b = sd[name + ".weight"], sd[name + ".bias"]
    if emulate_bf16:
        split = emulate_bf16 == "hr_split"
        if not (split and (name in _HR_SIDE or name == "conv_first")):
            w = _bf16r(w)
        if not (split and name == "conv_first"):
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
                          momentum: float = 0.1, eps: float = 1e-5) -> torch.Tensor:
    x = img
    layout = discriminator_layout(base)
    for j, (ci, _cin, _cout, stride, bi) in enumerate(layout):
        x = F.conv2d(x, sd[f"model.{ci}.weight"], sd[f"model.{ci}.bias"], stride=stride, padding=1)
        if j == len(layout) - 1:
            break
        if bi is not None:
            x = F.batch_norm(x, sd[f"model.{bi}.running_mean"], sd[f"model.{bi}.running_var"],
                             sd[f"model.{bi}.weight"], sd[f"model.{bi}.bias"],
                             training=training, momentum=momentum, eps=eps)
        x = F.leaky_relu(x, LRELU_SLOPE)
    return x


# --------------------------------------------------------------------------
# metrics used by the parity gates
# --------------------------------------------------------------------------
def psnr(a: torch.Tensor, b: torch.Tensor, peak: float = 1.0) -> float:
    mse = torch.mean((a.double() - b.double()) ** 2).item()
    return float("inf") if mse == 0 else 10.0 * math.log10(peak * peak / mse)


def inside_unit_fraction(t: torch.Tensor) -> float:
    return ((t > 0) & (t < 1)).double().mean().item()
