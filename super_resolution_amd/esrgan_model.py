"""Drop-in for the reference's ``esrgan/models/esrgan_model.py`` generator classes.

Same class names, constructor arguments, parameter names / shapes / init order (so the same
``torch.manual_seed`` gives the same weights and ``state_dict`` files interchange), same
call sites (``netG(lr)``, ``.clamp(0,1)``, ``nn.DataParallel``-prefixed checkpoints).  The
parameters stay fp32 OIHW ``nn.Conv2d`` tensors; the arithmetic runs in libsr355.so:

* ``GeneratorRRDB.forward``  -> one ``sr355::generator_forward`` call (351 fused conv launches)
  replacing /root/reference/esrgan/models/esrgan_model.py:55-62
* ``ResidualDenseBlock`` / ``RRDB`` forward -> ``sr355::conv3x3_fwd`` per conv, writing at channel
  offsets of one blocked buffer, replacing :18-24 and :35-39

There is no CPU path: tensors must live on a ROCm device.
"""
from __future__ import annotations

import operator
import warnings
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn
from torch import Tensor
from torch.autograd.function import once_differentiable

from . import fp32_train, ops
from ._lib import DTYPE_BF16, DTYPE_F32, HR_F16, TRUNK_RRDB, Sr355Error

_PRECISIONS = {"bf16": DTYPE_BF16, "fp32": DTYPE_F32, "f32": DTYPE_F32}
_VERSION_OF = operator.attrgetter("_version")
_SLOPE = 0.2
_RES = 0.2


def _conv(cin: int, cout: int) -> nn.Conv2d:
    # Parameter container with the reference's default initialisation (nn.Conv2d k=3, s=1, p=1).
    return nn.Conv2d(cin, cout, kernel_size=3, stride=1, padding=1)


def _is_replica(module: nn.Module) -> bool:
    """True for a per-device copy made by ``nn.parallel.replicate`` (``nn.DataParallel`` builds them anew in
    every forward)."""
    return bool(module.__dict__.get("_is_replica", False))


class _PackCache:
    """Packed copies of a module's conv parameters, rebuilt when any parameter changes.  One entry per (device,
    arithmetic): ``nn.DataParallel`` replicas share this object (``replicate`` copies a module's ``__dict__``
    shallowly) and call it from one thread per device, so an entry is read and replaced as a whole and a caller
    only ever returns the pack it looked up or built itself.

    Only a module that owns its parameters may cache.  A replica's parameters are fresh broadcast outputs in every
    forward: their version counters restart at 0 and the caching allocator hands out the same addresses again
    iteration after iteration, so (version, address) of a replica says nothing about the values behind it.  A
    replica (``cacheable=False``) therefore packs on every call and leaves the owner's entries alone."""

    def __init__(self) -> None:
        self.entries: Dict[Tuple, Tuple] = {}

    def get(self, params: List[Tensor], dtype, builder, cacheable: bool = True):
        if not cacheable:
            return builder()
        slot = (dtype, params[0].device)
        key = (tuple(map(_VERSION_OF, params)), tuple(map(Tensor.data_ptr, params)))
        entry = self.entries.get(slot)
        if entry is None or entry[0] != key:
            entry = (key, builder())
            self.entries[slot] = entry
        return entry[1]


class ResidualDenseBlock(nn.Module):
    """Five 3x3 convs over a growing channel stack; ``x + 0.2 * conv5(...)``."""

    def __init__(self, in_channels: int = 64, growth_channels: int = 32, *, precision: str = "bf16"):
        super().__init__()
        self.in_channels, self.growth_channels = in_channels, growth_channels
        for k in range(5):
            setattr(self, f"conv{k + 1}",
                    _conv(in_channels + k * growth_channels, growth_channels if k < 4 else in_channels))
        self.res_scale = _RES
        self.precision = precision
        self._cache = _PackCache()

    def _convs(self) -> List[nn.Conv2d]:
        return [getattr(self, f"conv{k}") for k in range(1, 6)]

    def _packed(self, dtype: int):
        params = [p for c in self._convs() for p in (c.weight, c.bias)]
        return self._cache.get(params, dtype, lambda: [ops.pack_weight(c.weight.detach().float(), c.bias.detach().float(), dtype)
                                                        for c in self._convs()], not _is_replica(self))

    def _run(self, buf: Tensor, trunk_in: Tensor, trunk_out: Tensor, nxt: Tensor, dtype: int,
             outer: Optional[Tensor] = None, outer_out: Optional[Tensor] = None) -> None:
        """``buf`` holds x in units [0, nf); growth outputs are appended in place; conv5 writes
        ``trunk_in + 0.2*conv5`` (optionally folded into ``outer + 0.2*(...)``) to the fp32 trunk
        and to units [0, nf) of ``nxt``."""
        cpu = ops.UNIT_CHANNELS[dtype]
        nf, gc = self.in_channels, self.growth_channels
        packed = self._packed(dtype)
        for k in range(4):
            wp, bp = packed[k]
            ops.conv3x3_fwd(buf, (nf + k * gc) // cpu, wp, bp, gc, _SLOPE, 0, None, 0.0, None, 0.0,
                            buf, (nf + k * gc) // cpu, None, None)
        wp, bp = packed[4]
        if outer is None:
            ops.conv3x3_fwd(buf, (nf + 4 * gc) // cpu, wp, bp, nf, 1.0, 0, trunk_in, self.res_scale, None, 0.0,
                            nxt, 0, trunk_out, None)
        else:
            ops.conv3x3_fwd(buf, (nf + 4 * gc) // cpu, wp, bp, nf, 1.0, 0, trunk_in, self.res_scale, outer, _RES,
                            nxt, 0, outer_out, None)

    def _packed_bwd(self):
        """Transposed (data-gradient) packs of the five convs, stacked as the gradient stack
        [dOut | dY4 | dY3 | dY2 | dY1] reads them; conv5's slices carry the dense skip's scale."""
        nf, gc = self.in_channels, self.growth_channels
        weights = [c.weight for c in self._convs()]

        def build():
            packs = []
            for j in (4, 3, 2, 1, 0):                      # step producing dY_j (j = 0: the block input's gradient)
                ci_lo, n_out = (nf + (j - 1) * gc, gc) if j else (0, nf)
                units = (nf + (4 - j) * gc) // 8
                wt = torch.zeros(units * 9 * ((n_out + 31) // 32 * 32) * 16, dtype=torch.uint8, device=weights[0].device)
                ops.pack_weight_transposed(weights[4].detach(), ci_lo, n_out, self.res_scale, wt, units, 0, DTYPE_BF16)
                for k in range(4, j, -1):
                    ops.pack_weight_transposed(weights[k - 1].detach(), ci_lo, n_out, 1.0, wt, units,
                                               (nf + (4 - k) * gc) // 8, DTYPE_BF16)
                packs.append(wt)
            return packs
        return self._cache.get(weights, ("bwd", DTYPE_BF16), build, not _is_replica(self))

    def _backward(self, act: Tensor, g: Tensor, outer_scale: float, grads: List[Tensor]) -> Tensor:
        """Backward of ``x + res_scale * conv5(...)`` given the saved block buffer ``act`` and ``g``, the gradient
        (NCHW fp32) of ``outer_scale`` times ... the block output: returns the conv path's share of the input
        gradient, ``outer_scale * d(conv path)/dx^T g``, and appends (dw, db) of conv1..conv5 to ``grads``."""
        nf, gc = self.in_channels, self.growth_channels
        N, H, _u, W, _c = act.shape
        dev = act.device
        stack = ops.to_blocked(g, act.shape[2], DTYPE_BF16)            # units [0, nf/8): dOut
        packs = self._packed_bwd()
        zero = torch.zeros(max(64, (max(nf, gc) + 31) // 32 * 32), dtype=torch.float32, device=dev)
        dxc = torch.empty((N, H, nf // 4, W, 4), dtype=torch.float32, device=dev)
        mask_off = [(nf + (3 - k) * gc) // 8 for k in range(4)]
        if W <= ops.DENSE_BLOCK_MAX_WIDTH and nf == 64 and gc == 32:
            ops.dense_block_fwd(stack, packs, [zero] * 5, 1.0, True, None, 1.0, None, 0.0, None, dxc, mask=act,
                                mask_unit_off=mask_off, mask_slope=_SLOPE)
        else:
            for k in range(4):
                ops.conv3x3_fwd(stack, (nf + k * gc) // 8, packs[k], zero, gc, 1.0, 0, None, 0.0, None, 0.0, stack,
                                (nf + k * gc) // 8, None, None, mask=act, mask_unit_off=mask_off[k], mask_slope=_SLOPE)
            ops.conv3x3_fwd(stack, (nf + 4 * gc) // 8, packs[4], zero, nf, 1.0, 0, None, 0.0, None, 0.0, None, 0, dxc, None)
        if outer_scale != 1.0:          # the stack was built from the unscaled gradient: everything is linear in it
            dxc = dxc * outer_scale
        convs = self._convs()
        dws = [torch.empty_like(c.weight, dtype=torch.float32) for c in convs]
        dbs = [torch.empty((gc if k < 4 else nf,), dtype=torch.float32, device=dev) for k in range(5)]
        if gc <= 32 and nf <= 64 and not ((nf | gc) & 7):
            # the five weight gradients as one batched launch + one reduction
            ops.dense_block_wgrad_now(act, stack, nf, gc, outer_scale, outer_scale * self.res_scale, dws, dbs)
        else:
            for k in range(5):
                dy_off = (nf + (3 - k) * gc) // 8 if k < 4 else 0
                ops.conv3x3_wgrad(act, nf + k * gc, 0, stack, dy_off, gc if k < 4 else nf,
                                  outer_scale * (1.0 if k < 4 else self.res_scale), 0, dws[k], dbs[k])
        for dw, db in zip(dws, dbs):
            grads += [dw, db]
        return ops.from_blocked(dxc, 0, nf)

    def forward(self, x: Tensor) -> Tensor:
        _check_input(x, self.in_channels)
        if x.shape[0] == 0:
            return _empty_result(x, self.in_channels, 1, list(self.parameters()))
        dtype = _PRECISIONS[self.precision]
        cast = self.conv1.weight.dtype
        if _records_graph(self, x):
            _need_fp32_parameters(cast)
            if dtype != DTYPE_BF16:        # exact-fp32 training: one autograd node per conv
                return fp32_train.block_forward(self, x.float(), False)
            return _BlockFunction.apply(x.float().contiguous(), self, *[p for c in self._convs() for p in (c.weight, c.bias)])
        cpu = ops.UNIT_CHANNELS[dtype]
        units = (self.in_channels + 4 * self.growth_channels) // cpu
        x = x.float()
        buf = ops.to_blocked(x, units, dtype)
        nxt = torch.empty_like(buf)
        trunk = ops.to_blocked(x, self.in_channels // 4, DTYPE_F32)
        self._run(buf, trunk, trunk, nxt, dtype)
        out = ops.from_blocked(trunk, 0, self.in_channels)
        return out if cast == torch.float32 else out.to(cast)


class RRDB(nn.Module):
    """Three dense blocks and the outer ``x + 0.2 * out`` skip."""

    def __init__(self, in_channels: int, growth_channels: int, *, precision: str = "bf16"):
        super().__init__()
        self.in_channels, self.growth_channels = in_channels, growth_channels
        for r in (1, 2, 3):
            setattr(self, f"rdb{r}", ResidualDenseBlock(in_channels, growth_channels, precision=precision))
        self.res_scale = _RES
        self.precision = precision

    def forward(self, x: Tensor) -> Tensor:
        _check_input(x, self.in_channels)
        if x.shape[0] == 0:
            return _empty_result(x, self.in_channels, 1, list(self.parameters()))
        dtype = _PRECISIONS[self.precision]
        cast = self.rdb1.conv1.weight.dtype
        if _records_graph(self, x):
            _need_fp32_parameters(cast)
            if dtype != DTYPE_BF16:
                return fp32_train.block_forward(self, x.float(), True)
            params = [p for r in (self.rdb1, self.rdb2, self.rdb3) for c in r._convs() for p in (c.weight, c.bias)]
            return _BlockFunction.apply(x.float().contiguous(), self, *params)
        cpu = ops.UNIT_CHANNELS[dtype]
        units = (self.in_channels + 4 * self.growth_channels) // cpu
        x = x.float()
        cur = ops.to_blocked(x, units, dtype)
        nxt = torch.empty_like(cur)
        outer = ops.to_blocked(x, self.in_channels // 4, DTYPE_F32)   # fp32 RRDB input
        trunk = torch.empty_like(outer)
        self.rdb1._run(cur, outer, trunk, nxt, dtype)
        self.rdb2._run(nxt, trunk, trunk, cur, dtype)
        self.rdb3._run(cur, trunk, None, nxt, dtype, outer=outer, outer_out=outer)
        out = ops.from_blocked(outer, 0, self.in_channels)
        return out if cast == torch.float32 else out.to(cast)


class GeneratorRRDB(nn.Module):
    """ESRGAN x4 generator.  ``precision`` selects the kernel arithmetic: ``"bf16"`` (bf16 storage,
    fp32 accumulation, fp32 residual trunk) or ``"fp32"`` (exact fp32 matrix-core path; it trains too, conv by conv
    through ``fp32_train``: the accuracy path, several times slower than the bf16 backward).
    ``hr_split`` (bf16 only) carries the bf16 rounding residue of the weights of the four
    HR-side convs, and of the input and weights of conv_first, as extra MFMA terms; their rounding
    otherwise dominates the output error (50.0 dB -> 53.9 dB PSNR against fp32 at +8 % MFMA work).
    ``hr_f16`` (bf16 only, default): forwards that record nothing for a backward pass (``torch.no_grad()``, frozen
    parameters) run the four HR-side convs in IEEE fp16 instead — weights and the activations those convs read keep
    11 significant bits instead of 8 at the same MFMA rate and without the second weight image: 58.6 dB against fp32
    and 6 % less time than ``hr_split``.  fp16 saturates at 65504: trained ESRGAN activations are O(1-10) there, and
    the reference itself runs fp16 under autocast, but pass ``False`` for weights with extreme HR-side activations.
    Forwards recorded for backward always use the ``hr_split`` form (the backward kernels read bf16 activations),
    so a grad-mode output and a no-grad output of the same input differ by that rounding (about 1e-3).
    ``dense_skip_fp32`` (bf16 only): ``False`` (default) lets no-grad forwards add the dense-block skips onto the bf16
    activations themselves and keep fp32 carriers only for the RRDB and global skips — 41 % fewer epilogue bytes per
    RRDB, about 3 dB of PSNR (55.6 dB with ``hr_f16``).  ``True`` keeps an fp32 carrier for every skip (58.6 dB);
    recorded forwards always do.
    ``direct_grads``: the backward pass writes (or adds) every parameter's ``.grad`` itself, as slices of one
    flat buffer, instead of handing 702 gradients to autograd's per-parameter accumulation: the same ``.grad``
    fields after ``loss.backward()`` (optimizers, ``GradScaler``, gradient accumulation see no difference) at a
    third of the host time of a small-patch step.  Pass ``False`` to keep per-parameter autograd edges
    (``torch.autograd.grad(loss, netG.parameters())``, tensor hooks on the parameters).
    ``checked``: a no-grad forward of a large batch / image runs its body as one persistent launch whose workgroups
    must all be resident on the GPU; if another process holds part of the GPU its waits give up and the output is
    invalid, which an unchecked forward reports only on the NEXT call (or through ``ops.chain_status()``).  With
    ``checked=True`` every such forward waits for its launch, and one that gave up is run again as one launch per conv
    (a warning says so): the call returns a valid result at the price of one host synchronisation per forward."""

    def __init__(self, in_channels: int = 3, out_channels: int = 3, num_features: int = 64,
                 num_blocks: int = 23, growth_channels: int = 32, *, precision: str = "bf16",
                 hr_split: bool = True, direct_grads: bool = True, hr_f16: bool = True,
                 dense_skip_fp32: bool = False, checked: bool = False):
        super().__init__()
        self.checked = bool(checked)
        self.hr_split = bool(hr_split)
        self.hr_f16 = bool(hr_f16)
        self.dense_skip_fp32 = bool(dense_skip_fp32)
        self.direct_grads = bool(direct_grads)
        if precision not in _PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_PRECISIONS)}")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.num_features, self.num_blocks, self.growth_channels = num_features, num_blocks, growth_channels
        self.precision = precision
        self.conv_first = _conv(in_channels, num_features)
        self.body = nn.Sequential(*[RRDB(num_features, growth_channels, precision=precision)
                                    for _ in range(num_blocks)])
        for name in ("trunk_conv", "upconv1", "upconv2", "HR_conv"):
            setattr(self, name, _conv(num_features, num_features))
        self.conv_last = _conv(num_features, out_channels)
        self._cache = _PackCache()
        self._cache_inf = _PackCache()
        self._cache_bwd = _PackCache()
        self._workspace: Optional[Dict] = None

    # -- parameter plumbing ---------------------------------------------------------------
    def conv_modules(self) -> List[nn.Conv2d]:
        """The convs in state_dict order.  Walked on every call (a submodule may have been replaced) through the
        ``_modules`` dicts: ``nn.Module.__getattr__`` on 400 names costs more than the rest of a training step's
        host work on this module."""
        top = self._modules
        mods = [top["conv_first"]]
        for rrdb in top["body"]._modules.values():
            blocks = rrdb._modules
            for name in ("rdb1", "rdb2", "rdb3"):
                convs = blocks[name]._modules
                mods.append(convs["conv1"])
                mods.append(convs["conv2"])
                mods.append(convs["conv3"])
                mods.append(convs["conv4"])
                mods.append(convs["conv5"])
        mods += [top["trunk_conv"], top["upconv1"], top["upconv2"], top["HR_conv"], top["conv_last"]]
        return mods

    def conv_parameters(self) -> List[Tensor]:
        """weight, bias of every conv in state_dict order, read from the modules on every call (a parameter may
        have been replaced since the last one).  An ``nn.DataParallel`` replica holds them as plain attributes."""
        out = []
        for c in self.conv_modules():
            d = c._parameters
            w, b = d.get("weight"), d.get("bias")
            out.append(w if w is not None else c.weight)
            out.append(b if b is not None else c.bias)
        return out

    def cfg(self, inference: bool = False) -> List[int]:
        """The C-side configuration; ``inference`` selects the fp16 HR side where the module allows it."""
        dtype = _PRECISIONS[self.precision]
        hr, trunk = 0, 1
        if dtype == DTYPE_BF16:
            hr = HR_F16 if (inference and self.hr_f16) else int(self.hr_split)
            trunk = TRUNK_RRDB if (inference and not self.dense_skip_fp32) else 1
        return [self.in_channels, self.out_channels, self.num_features, self.num_blocks,
                self.growth_channels, dtype, hr, trunk]

    def packed_parameters(self, inference: bool = False) -> Tensor:
        params = self.conv_parameters()
        cfg = self.cfg(inference)
        cache = self._cache_inf if cfg[6] == HR_F16 else self._cache
        def build():
            if params[0].dtype == torch.float32:      # (no per-tensor calls on the usual path: 702 tensors a repack)
                return ops.generator_pack_now(params[0::2], params[1::2], cfg)
            # a module cast by .half() / .bfloat16() / .double() packs from fp32 copies of its parameters
            return ops.generator_pack_now([w.detach().float() for w in params[0::2]],
                                          [b.detach().float() for b in params[1::2]], cfg)
        return cache.get(params, (cfg[5], cfg[6]), build, not _is_replica(self))

    def workspace(self, N: int, H: int, W: int, device) -> Tensor:
        """Scratch memory of a no-grad forward, kept per device and grown on demand."""
        need = ops.generator_workspace_bytes(self.cfg(), N, H, W)
        held = self._workspace
        if held is None:
            held = self._workspace = {}
        ws = held.get(device)
        if ws is None or ws.numel() < need:
            held.pop(device, None)          # release before growing
            ws = None
            ws = torch.empty(need, dtype=torch.uint8, device=device)
            held[device] = ws
        return ws

    def packed_bwd_parameters(self) -> Tensor:
        weights = self.conv_parameters()[0::2]
        cfg = self.cfg()
        return self._cache_bwd.get(weights, (cfg[5], cfg[6]), lambda: ops.generator_pack_bwd_now(weights, cfg),
                                   not _is_replica(self))

    # -- forward --------------------------------------------------------------------------
    def forward(self, x: Tensor) -> Tensor:
        _check_input(x, self.in_channels)
        if x.shape[0] == 0:
            return _empty_result(x, self.out_channels, 4, list(self.parameters()))
        xin = x.float().contiguous()
        params = self.conv_parameters()
        cast = params[0].dtype          # a cast module (netG.half()) returns its own dtype, as nn.Conv2d does
        if torch.is_grad_enabled() and (xin.requires_grad or any(p.requires_grad for p in params)):
            _need_fp32_parameters(cast)
            if _PRECISIONS[self.precision] != DTYPE_BF16:
                # exact-fp32 training (the reference without autocast): one autograd node per conv, gradients
                # through autograd's own accumulation
                return fp32_train.generator_forward(self, xin)
            return _GeneratorFunction.apply(xin, self, None, *_graph_params(self, params))
        N, _c, H, W = xin.shape
        out = self._checked(xin.device, lambda: ops.generator_forward(
            xin.detach(), self.packed_parameters(True), self.workspace(N, H, W, xin.device), self.cfg(True)))
        return out if cast == torch.float32 else out.to(cast)

    def _checked(self, device, run):
        """``run()``; with ``checked`` also wait for the body's persistent launch and, if it gave up, run again (the
        library has switched the device to one launch per conv by then)."""
        out = run()
        if getattr(self, "checked", False):
            try:
                ops.chain_status(True, device)
            except Sr355Error as exc:
                warnings.warn(f"{exc}; the forward was repeated as one launch per conv")
                out = run()
        return out

    @torch.no_grad()
    def upscale_u8(self, image: Tensor) -> Tensor:
        """Image in, image out: uint8 ``[H, W, 3]`` or ``[N, H, W, 3]`` (the layout of ``numpy.asarray(PIL image)``)
        on the module's device -> uint8 ``[4H, 4W, 3]`` / ``[N, 4H, 4W, 3]``.  Bit-identical to
        ``ToPILImage()(self(ToTensor()(image)).clamp(0, 1))`` of /root/reference/esrgan/test.py:69-73, with the
        byte -> float conversion folded into the first layout kernel and clamp / scale / truncate into conv_last's
        store, so neither fp32 image exists in memory."""
        single = image.dim() == 3
        img = image.unsqueeze(0) if single else image
        if img.dim() != 4 or img.dtype != torch.uint8 or img.shape[-1] != self.in_channels:
            raise RuntimeError(f"expected a uint8 HWC image with {self.in_channels} channels, got "
                               f"{tuple(image.shape)} {image.dtype}")
        if not img.is_cuda:
            raise RuntimeError("super_resolution_amd runs on ROCm devices only: move the image to 'cuda' "
                               "(there is no CPU fallback)")
        N, H, W, _c = img.shape
        if N == 0:
            return img.new_zeros((0, 4 * H, 4 * W, self.out_channels))
        if H == 0 or W == 0:
            raise RuntimeError(f"an image without pixels: {tuple(image.shape)}")
        out = self._checked(img.device, lambda: ops.generator_forward_u8(
            img.contiguous(), self.packed_parameters(True), self.workspace(N, H, W, img.device), self.cfg(True)))
        return out[0] if single else out


def _writes_grads_itself(module: "GeneratorRRDB", params: List[Tensor]) -> bool:
    """``direct_grads`` applies to a module that owns its parameters.  In an ``nn.DataParallel`` replica they are
    outputs of a broadcast (not leaves): their gradients have to flow back through autograd to the originals."""
    return bool(module.direct_grads and params and params[0].is_leaf and params[-1].is_leaf)


def _graph_params(module: "GeneratorRRDB", params: List[Tensor]) -> List[Tensor]:
    """The parameters the autograd node takes as inputs: all of them, or with ``direct_grads`` one trainable
    parameter that merely ties the node into the graph (the backward then fills ``.grad`` itself)."""
    if not _writes_grads_itself(module, params):
        return params
    for p in params:
        if p.requires_grad:
            return [p]
    return []


class _GeneratorFunction(torch.autograd.Function):
    """Autograd node of the whole generator: the forward keeps the activations in one buffer, the
    backward is one ``sr355::generator_backward`` call that fills every parameter gradient
    (what ``loss.backward()`` does over the recorded graph in esrgan/train.py:161, :220)."""

    @staticmethod
    def forward(ctx, x: Tensor, module: "GeneratorRRDB", reducer, *params: Tensor) -> Tensor:
        cfg = module.cfg()
        N, _c, H, W = (int(v) for v in x.shape)
        saved = torch.empty(ops.generator_saved_bytes(cfg, N, H, W), dtype=torch.uint8, device=x.device)
        ws = torch.empty(ops.generator_train_workspace_bytes(cfg, N, H, W), dtype=torch.uint8, device=x.device)
        out = ops.generator_forward_train(x.detach(), module.packed_parameters(), saved, ws, cfg)
        ctx.module, ctx.saved, ctx.lr_shape, ctx.reducer = module, saved, (N, H, W), reducer
        ctx.need_x = x.requires_grad
        ctx.n_graph_params = len(params)
        ctx.direct = _writes_grads_itself(module, module.conv_parameters())
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out: Tensor):
        module, cfg = ctx.module, ctx.module.cfg()
        N, H, W = ctx.lr_shape
        ws = torch.empty(ops.generator_bwd_workspace_bytes(cfg, N, H, W), dtype=torch.uint8, device=grad_out.device)
        gout = grad_out.float().contiguous()
        if ctx.reducer is None:
            flat, gin = ops.generator_backward(gout, module.packed_bwd_parameters(), ctx.saved, ws, cfg, H, W, ctx.need_x)
        else:
            # data parallel: phase groups, each followed by the asynchronous all-reduce of its gradient slice
            from .ddp import reduce_in_buckets
            flat = torch.empty(ops.generator_grads_floats(cfg), dtype=torch.float32, device=gout.device)
            gin = (torch.empty((N, module.in_channels, H, W), dtype=torch.float32, device=gout.device)
                   if ctx.need_x else None)
            packed_bwd = module.packed_bwd_parameters()
            reduce_in_buckets(flat, ctx.reducer.buckets,
                              lambda pb, pe: ops.generator_backward_range(gout, packed_bwd, ctx.saved, ws, flat, gin, cfg,
                                                                          H, W, pb, pe), ctx.reducer.group,
                              getattr(ctx.reducer, "algorithm", "all_reduce"))
        params = module.conv_parameters()
        # one split for all 702 slices, then a reshape of the 4-d ones (biases are 1-d already)
        views = list(flat.split_with_sizes([p.numel() for p in params]))
        for i in range(0, len(views), 2):
            views[i] = views[i].view(params[i].shape)
        gx = gin if ctx.need_x else None
        if not ctx.direct:
            return (gx, None, None, *views)
        # what autograd's accumulation would do, parameter by parameter: the first gradient becomes .grad
        # (a slice of the flat buffer), later ones are added to it
        have_p, have_v = [], []
        for p, v in zip(params, views):
            if not p.requires_grad:
                continue
            if p.grad is None:
                p.grad = v
            else:
                have_p.append(p.grad)
                have_v.append(v)
        if have_p:
            torch._foreach_add_(have_p, have_v)
        return (gx, None, None, *([None] * ctx.n_graph_params))


# RRDBNet is the name BasicSR-derived code uses for the same network.
RRDBNet = GeneratorRRDB


def _check_input(x: Tensor, channels: int) -> None:
    if x.dim() != 4 or x.shape[1] != channels:
        raise RuntimeError(f"expected an NCHW tensor with {channels} channels, got {tuple(x.shape)}")
    if not x.is_cuda:
        raise RuntimeError("super_resolution_amd runs on ROCm devices only: move the module and its input to "
                           "'cuda' (there is no CPU fallback)")
    if not x.is_floating_point():
        # nn.Conv2d refuses it the same way ("Input type ... and weight type ... should be the same")
        raise RuntimeError(f"Input type ({x.dtype}) and weight type (torch.float32) should be the same: pass a "
                           "floating-point tensor (images as bytes go through upscale_u8)")
    if x.shape[0] and (x.shape[2] == 0 or x.shape[3] == 0):
        raise RuntimeError(f"Kernel size (3 x 3) can't be greater than the padded input size "
                           f"({x.shape[2] + 2} x {x.shape[3] + 2})")


def _need_fp32_parameters(dtype) -> None:
    if dtype != torch.float32:
        raise RuntimeError(f"a recorded forward needs fp32 parameters (the backward writes fp32 gradients), got {dtype}: "
                           "keep the module in fp32 -- the kernels compute in bf16 either way -- or call it under "
                           "torch.no_grad()")


def _empty_result(x: Tensor, channels: int, scale: int, params) -> Tensor:
    """What the reference's convs return for a batch of no images: ``[0, channels, scale H, scale W]``, connected to the
    graph (all-zero gradients) where one is recorded."""
    out = x.new_zeros((0, channels, scale * x.shape[2], scale * x.shape[3]), dtype=torch.float32)
    if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params)):
        out = out + x.float().sum() * 0 + sum(p.sum() for p in params if p.requires_grad) * 0
    return out


def _records_graph(module: nn.Module, x: Tensor) -> bool:
    return torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in module.parameters()))


class _BlockFunction(torch.autograd.Function):
    """Autograd node of a stand-alone ``ResidualDenseBlock`` or ``RRDB``
    (/root/reference/esrgan/models/esrgan_model.py:18-24, :35-39): the forward keeps one blocked buffer per dense
    block, the backward runs each block's five data gradients (one launch for images up to 32 pixels wide) and its
    five weight gradients on the same kernels the generator's backward uses."""

    @staticmethod
    def forward(ctx, x: Tensor, module, *params: Tensor) -> Tensor:
        blocks = [module] if isinstance(module, ResidualDenseBlock) else [module.rdb1, module.rdb2, module.rdb3]
        nf, gc = blocks[0].in_channels, blocks[0].growth_channels
        units = (nf + 4 * gc) // 8
        xd = x.detach()
        bufs = [ops.to_blocked(xd, units, DTYPE_BF16)] + [torch.empty((x.shape[0], x.shape[2], units, x.shape[3], 8),
                                                                       dtype=torch.bfloat16, device=x.device)
                                                           for _ in blocks]
        outer = ops.to_blocked(xd, nf // 4, DTYPE_F32)            # fp32 input (the RRDB skip reads it)
        trunk = torch.empty_like(outer)
        if len(blocks) == 1:
            blocks[0]._run(bufs[0], outer, trunk, bufs[1], DTYPE_BF16)
            out = trunk
        else:
            blocks[0]._run(bufs[0], outer, trunk, bufs[1], DTYPE_BF16)
            blocks[1]._run(bufs[1], trunk, trunk, bufs[2], DTYPE_BF16)
            blocks[2]._run(bufs[2], trunk, None, bufs[3], DTYPE_BF16, outer=outer, outer_out=outer)
            out = outer
        ctx.module, ctx.blocks, ctx.bufs = module, blocks, bufs[:len(blocks)]
        ctx.need_x = x.requires_grad
        return ops.from_blocked(out, 0, nf)

    @staticmethod
    @once_differentiable
    def backward(ctx, g: Tensor):
        g = g.float().contiguous()
        blocks, bufs = ctx.blocks, ctx.bufs
        if len(blocks) == 1:
            grads: List[Tensor] = []
            dx = g + blocks[0]._backward(bufs[0], g, 1.0, grads)
        else:
            # out = x + 0.2 * rdb3(rdb2(rdb1(x))): d3 = 0.2 g + 0.2 J3^T g, d2 = d3 + J2^T d3, d1 = d2 + J1^T d2
            s = ctx.module.res_scale
            g3, g2, g1 = [], [], []
            d3 = s * g + blocks[2]._backward(bufs[2], g, s, g3)
            d2 = d3 + blocks[1]._backward(bufs[1], d3, 1.0, g2)
            d1 = d2 + blocks[0]._backward(bufs[0], d2, 1.0, g1)
            dx = g + d1
            grads = g1 + g2 + g3
        return (dx if ctx.need_x else None, None, *grads)
