"""``torch.library`` custom ops (namespace ``sr355``) over the C ABI of libsr355.so.

PyTorch supplies device memory and the current HIP stream; all arithmetic runs in the
HIP kernels.  Blocked activation tensors have shape ``[N, H, U, W, c]`` with ``c`` = 8
(bf16) or 4 (fp32) channels per 16-byte unit; fp32 trunk tensors are ``[N, H, C/4, W, 4]``.
"""
from __future__ import annotations

import contextlib
import ctypes as C
from typing import List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from . import _lib
from ._lib import DTYPE_BF16, DTYPE_F32, ConvArgs, DenseBlockArgs, DenseWgradArgs, GeneratorCfg, WgradArgs, check

_TORCH_DTYPE = {DTYPE_BF16: torch.bfloat16, DTYPE_F32: torch.float32}
_CODE = {torch.bfloat16: DTYPE_BF16, torch.float32: DTYPE_F32}
UNIT_CHANNELS = {DTYPE_BF16: 8, DTYPE_F32: 4}


# torch keeps the raw-handle getters private; where a torch version drops them the public (slower) API takes over
_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_CUR_DEVICE = getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device


def _stream() -> C.c_void_p:
    """The current HIP stream of the current device, as the C ABI takes it (the raw handle, without building a
    ``torch.cuda.Stream`` object around it: this runs once per kernel call)."""
    if _RAW_STREAM is not None:
        return C.c_void_p(_RAW_STREAM(_CUR_DEVICE()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


_SAME_DEVICE = contextlib.nullcontext()


def _on(device: torch.device):
    """Context in which ``device`` is the current device; nothing to enter when it already is."""
    idx = device.index
    return _SAME_DEVICE if idx is None or idx == _CUR_DEVICE() else torch.cuda.device(device)


def _ptr(t: Optional[Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _need_cuda(*ts: Optional[Tensor]) -> None:
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("sr355 ops run on the GPU only (got a CPU tensor); there is no CPU fallback")
        if not t.is_contiguous():
            raise RuntimeError("sr355 ops need contiguous tensors")


def _nbytes(t: Optional[Tensor]) -> int:
    return 0 if t is None else t.numel() * t.element_size()


# ------------------------------------------------------------------------------------
@torch.library.custom_op("sr355::to_blocked", mutates_args=())
def to_blocked(x: Tensor, units: int, dtype: int, split: int = 0) -> Tensor:
    """NCHW fp32 -> blocked [N,H,units,W,c]; channels beyond C are zero (split=1: channels [C,2C)
    carry the bf16 rounding residue of channels [0,C))."""
    _need_cuda(x)
    if x.dtype != torch.float32 or x.dim() != 4:
        raise RuntimeError("to_blocked expects an fp32 NCHW tensor")
    N, Cc, H, W = x.shape
    out = torch.empty((N, H, units, W, UNIT_CHANNELS[dtype]), dtype=_TORCH_DTYPE[dtype], device=x.device)
    with _on(x.device):
        check(_lib.lib().sr355_nchw_to_blocked(x.data_ptr(), N, Cc, H, W, dtype, out.data_ptr(), units, split,
                                               _nbytes(out), _stream()), "sr355_nchw_to_blocked")
    return out


@to_blocked.register_fake
def _(x, units, dtype, split=0):
    N, _c, H, W = x.shape
    return x.new_empty((N, H, units, W, UNIT_CHANNELS[dtype]), dtype=_TORCH_DTYPE[dtype])


@torch.library.custom_op("sr355::from_blocked", mutates_args=())
def from_blocked(buf: Tensor, unit_off: int, channels: int) -> Tensor:
    """Blocked units [unit_off, ...) -> NCHW fp32 with ``channels`` channels."""
    _need_cuda(buf)
    N, H, U, W, _c = buf.shape
    out = torch.empty((N, channels, H, W), dtype=torch.float32, device=buf.device)
    with _on(buf.device):
        check(_lib.lib().sr355_blocked_to_nchw(buf.data_ptr(), U, unit_off, _CODE[buf.dtype], N, channels, H, W,
                                               out.data_ptr(), _nbytes(out), _stream()), "sr355_blocked_to_nchw")
    return out


@from_blocked.register_fake
def _(buf, unit_off, channels):
    N, H, _u, W, _c = buf.shape
    return buf.new_empty((N, channels, H, W), dtype=torch.float32)


@torch.library.custom_op("sr355::pack_weight", mutates_args=())
def pack_weight(w: Tensor, bias: Tensor, dtype: int, split: int = 0) -> Tuple[Tensor, Tensor]:
    """OIHW fp32 weight + bias -> (packed weight bytes, zero-padded fp32 bias).  ``split`` is one of
    the SR355_SPLIT_* modes of include/sr355.h (bf16 only)."""
    _need_cuda(w, bias)
    if w.dtype != torch.float32 or w.dim() != 4 or w.shape[2:] != (3, 3):
        raise RuntimeError("pack_weight expects an fp32 [cout, cin, 3, 3] tensor")
    cout, cin = int(w.shape[0]), int(w.shape[1])
    L = _lib.lib()
    nbytes = L.sr355_packed_weight_bytes(cout, cin, dtype, split)
    if nbytes == 0:
        check(-1, "sr355_packed_weight_bytes")
    wp = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    bp = torch.empty((cout + 31) // 32 * 32, dtype=torch.float32, device=w.device)
    with _on(w.device):
        check(L.sr355_pack_weight(w.data_ptr(), bias.data_ptr(), cout, cin, dtype, split, wp.data_ptr(), nbytes,
                                  bp.data_ptr(), _stream()), "sr355_pack_weight")
    return wp, bp


@pack_weight.register_fake
def _(w, bias, dtype, split=0):
    cout, cin = w.shape[0], w.shape[1]
    cpu = UNIT_CHANNELS[dtype]
    cu = (((2 * cin if split == 2 else cin) + cpu - 1) // cpu + 1) // 2 * 2 * (2 if split in (1, 2) else 1)
    cp = (cout + 31) // 32 * 32
    return (w.new_empty(cu * 9 * cp * 16, dtype=torch.uint8), w.new_empty(cp, dtype=torch.float32))


def conv_args(in_buf: Tensor, cin_units: int, w_packed: Tensor, bias_packed: Tensor, cout: int,
              lrelu_slope: float, upsample: int,
              res1: Optional[Tensor], res1_scale: float, res2: Optional[Tensor], res2_scale: float,
              out_act: Optional[Tensor], out_unit_off: int,
              out_f32: Optional[Tensor], out_nchw: Optional[Tensor], weight_split: int = 0,
              mask: Optional[Tensor] = None, mask_unit_off: int = 0, mask_slope: float = 1.0,
              res1_weight: float = 1.0, res2_weight: float = 1.0, f16_in: bool = False,
              f16_out: bool = False, res_act: Optional[Tensor] = None, stride: int = 1,
              zero_insert: bool = False) -> ConvArgs:
    """The sr355_conv_args record for one conv launch (the tensors must outlive its use).  ``stride=2``: the output
    tensors have ``(in+1)//2`` rows and columns; ``zero_insert`` (with ``upsample=1``): the input is read as its
    zero-inserted x2 image, the data gradient of a stride-2 conv."""
    _need_cuda(in_buf, w_packed, bias_packed, res1, res2, out_act, out_f32, out_nchw, mask)
    a = ConvArgs()
    a.dtype = _CODE[in_buf.dtype]
    if out_act is not None:
        N, H, _U, W, _c = out_act.shape
    elif out_f32 is not None:
        N, H, _U, W, _c = out_f32.shape
    elif out_nchw is not None:
        N, _cc, H, W = out_nchw.shape
    else:
        raise RuntimeError("conv3x3_fwd: no output tensor given")
    a.N, a.H, a.W, a.upsample = int(N), int(H), int(W), int(upsample)
    a.in_dev, a.in_bytes = in_buf.data_ptr(), _nbytes(in_buf)
    a.in_units, a.cin_units = int(in_buf.shape[2]), int(cin_units)
    if stride == 2:
        want_h, want_w = int(in_buf.shape[1]), int(in_buf.shape[3])
        if (want_h + 1) // 2 != H or (want_w + 1) // 2 != W:
            raise RuntimeError(f"conv3x3_fwd: stride-2 output of a {want_h}x{want_w} input is "
                               f"{(want_h + 1) // 2}x{(want_w + 1) // 2}, not {H}x{W}")
        a.stride, a.in_H, a.in_W = 2, want_h, want_w
    else:
        z = 1 if zero_insert else 0
        want_h, want_w = (H + z) >> upsample, (W + z) >> upsample
        a.zero_insert = z
    if in_buf.shape[0] != N or in_buf.shape[1] != want_h or in_buf.shape[3] != want_w:
        raise RuntimeError(f"conv3x3_fwd: input {tuple(in_buf.shape)} does not match output N={N} H={H} W={W} "
                           f"upsample={upsample} stride={stride}")
    a.w_packed_dev, a.w_packed_bytes = w_packed.data_ptr(), _nbytes(w_packed)
    a.weight_split = int(weight_split)
    a.bias_packed_dev = bias_packed.data_ptr()
    a.cout, a.lrelu_slope = int(cout), float(lrelu_slope)
    a.res1_dev, a.res1_weight, a.res1_scale = _ptr(res1), float(res1_weight), float(res1_scale)
    a.res2_dev, a.res2_weight, a.res2_scale = _ptr(res2), float(res2_weight), float(res2_scale)
    if mask is not None:
        a.mask_dev, a.mask_bytes = mask.data_ptr(), _nbytes(mask)
        a.mask_units, a.mask_unit_off, a.mask_slope = int(mask.shape[2]), int(mask_unit_off), float(mask_slope)
    sizes = [_nbytes(t) for t in (res1, res2, out_f32) if t is not None]
    a.res_bytes = min(sizes) if sizes else 0
    a.out_act_dev, a.out_act_bytes = _ptr(out_act), _nbytes(out_act)
    a.out_units = int(out_act.shape[2]) if out_act is not None else 0
    a.out_unit_off = int(out_unit_off)
    a.out_f32_dev = _ptr(out_f32)
    a.out_nchw_dev, a.out_nchw_bytes = _ptr(out_nchw), _nbytes(out_nchw)
    a.f16_in, a.f16_out = int(bool(f16_in)), int(bool(f16_out))
    if res_act is not None:
        _need_cuda(res_act)
        a.res_act_dev, a.res_act_bytes, a.res_act_units = res_act.data_ptr(), _nbytes(res_act), int(res_act.shape[2])
    return a


def launch_conv(a: ConvArgs) -> None:
    """sr355_conv3x3_fwd on the current device's current stream."""
    check(_lib.lib().sr355_conv3x3_fwd(C.byref(a), _stream()), "sr355_conv3x3_fwd")


@torch.library.custom_op("sr355::conv3x3_fwd", mutates_args=("out_act", "out_f32", "out_nchw"))
def conv3x3_fwd(in_buf: Tensor, cin_units: int, w_packed: Tensor, bias_packed: Tensor, cout: int,
                lrelu_slope: float, upsample: int,
                res1: Optional[Tensor], res1_scale: float, res2: Optional[Tensor], res2_scale: float,
                out_act: Optional[Tensor], out_unit_off: int,
                out_f32: Optional[Tensor], out_nchw: Optional[Tensor], weight_split: int = 0,
                mask: Optional[Tensor] = None, mask_unit_off: int = 0, mask_slope: float = 1.0,
                res1_weight: float = 1.0, res2_weight: float = 1.0) -> None:
    """3x3/s1/p1 conv over units [0, cin_units) of ``in_buf`` with fused bias, LeakyReLU, sign mask,
    weighted skips and optional nearest-x2 upsample of the input; writes any subset of the outputs."""
    a = conv_args(in_buf, cin_units, w_packed, bias_packed, cout, lrelu_slope, upsample, res1, res1_scale,
                  res2, res2_scale, out_act, out_unit_off, out_f32, out_nchw, weight_split, mask, mask_unit_off,
                  mask_slope, res1_weight, res2_weight)
    with _on(in_buf.device):
        launch_conv(a)


@conv3x3_fwd.register_fake
def _(in_buf, cin_units, w_packed, bias_packed, cout, lrelu_slope, upsample, res1, res1_scale, res2, res2_scale,
      out_act, out_unit_off, out_f32, out_nchw, weight_split=0, mask=None, mask_unit_off=0, mask_slope=1.0,
      res1_weight=1.0, res2_weight=1.0):
    return None


DENSE_BLOCK_MAX_WIDTH = 32


def dense_block_args(buf: Tensor, w_packed: List[Tensor], bias_packed: List[Tensor], lrelu_slope: float,
                     store_growth: bool, res1: Optional[Tensor], res1_scale: float, res2: Optional[Tensor],
                     res2_scale: float, out_act: Optional[Tensor], out_f32: Optional[Tensor],
                     mask: Optional[Tensor] = None, mask_unit_off: Optional[List[int]] = None, mask_slope: float = 1.0,
                     res1_weight: float = 1.0, res2_weight: float = 1.0,
                     res_act: Optional[Tensor] = None) -> DenseBlockArgs:
    """The sr355_dense_block_args record of one fused dense-block launch (the tensors must outlive its use)."""
    _need_cuda(buf, res1, res2, out_act, out_f32, mask, res_act, *w_packed, *bias_packed)
    if buf.dtype != torch.bfloat16 or buf.dim() != 5 or buf.shape[2] != 24 or len(w_packed) != 5 or len(bias_packed) != 5:
        raise RuntimeError("dense_block_fwd: expected a blocked bf16 buffer [N, H, 24, W, 8] and five packed convs")
    a = DenseBlockArgs()
    a.N, a.H, a.W = int(buf.shape[0]), int(buf.shape[1]), int(buf.shape[3])
    a.buf_dev, a.buf_bytes, a.store_growth = buf.data_ptr(), _nbytes(buf), int(bool(store_growth))
    for k in range(5):
        a.w_packed_dev[k], a.w_packed_bytes[k] = w_packed[k].data_ptr(), _nbytes(w_packed[k])
        a.bias_packed_dev[k] = bias_packed[k].data_ptr()
    a.lrelu_slope = float(lrelu_slope)
    if mask is not None:
        a.mask_dev, a.mask_bytes, a.mask_units = mask.data_ptr(), _nbytes(mask), int(mask.shape[2])
        for k in range(4):
            a.mask_unit_off[k] = int(mask_unit_off[k])
    a.mask_slope = float(mask_slope)
    a.res1_dev, a.res1_weight, a.res1_scale = _ptr(res1), float(res1_weight), float(res1_scale)
    a.res2_dev, a.res2_weight, a.res2_scale = _ptr(res2), float(res2_weight), float(res2_scale)
    sizes = [_nbytes(t) for t in (res1, res2, out_f32) if t is not None]
    a.res_bytes = min(sizes) if sizes else 0
    if res_act is not None:
        a.res_act_dev, a.res_act_bytes, a.res_act_units = res_act.data_ptr(), _nbytes(res_act), int(res_act.shape[2])
    a.out_act_dev, a.out_act_bytes = _ptr(out_act), _nbytes(out_act)
    a.out_units = int(out_act.shape[2]) if out_act is not None else 0
    a.out_f32_dev = _ptr(out_f32)
    return a


def launch_dense_block(a: DenseBlockArgs) -> None:
    """sr355_dense_block_fwd on the current device's current stream."""
    check(_lib.lib().sr355_dense_block_fwd(C.byref(a), _stream()), "sr355_dense_block_fwd")


def launch_dense_wgrad(a: DenseWgradArgs) -> None:
    """sr355_dense_block_wgrad on the current device's current stream."""
    check(_lib.lib().sr355_dense_block_wgrad(C.byref(a), _stream()), "sr355_dense_block_wgrad")


DENSE_WGRAD_MAX_BLOCKS = 9      # SR355_DENSE_WGRAD_MAX_BLOCKS of include/sr355.h


def launch_dense_wgrad_blocks(blocks: Sequence[DenseWgradArgs]) -> None:
    """sr355_dense_blocks_wgrad: the weight gradients of up to nine dense blocks of one shape (an RRDB's, or two or three RRDBs')
    as one launch and one reduction; ``blocks[0].workspace`` must hold ``len(blocks)`` times the one-block size."""
    if not 1 <= len(blocks) <= DENSE_WGRAD_MAX_BLOCKS:
        raise RuntimeError(f"dense_blocks_wgrad takes 1 to {DENSE_WGRAD_MAX_BLOCKS} blocks")
    arr = (DenseWgradArgs * len(blocks))(*blocks)
    check(_lib.lib().sr355_dense_blocks_wgrad(arr, len(blocks), _stream()), "sr355_dense_blocks_wgrad")


@torch.library.custom_op("sr355::dense_block_fwd", mutates_args=("buf", "out_act", "out_f32"))
def dense_block_fwd(buf: Tensor, w_packed: List[Tensor], bias_packed: List[Tensor], lrelu_slope: float,
                    store_growth: bool, res1: Optional[Tensor], res1_scale: float, res2: Optional[Tensor],
                    res2_scale: float, out_act: Optional[Tensor], out_f32: Optional[Tensor],
                    mask: Optional[Tensor] = None, mask_unit_off: Optional[List[int]] = None, mask_slope: float = 1.0,
                    res1_weight: float = 1.0, res2_weight: float = 1.0, res_act: Optional[Tensor] = None) -> None:
    """A whole dense block (conv1..conv4 + LeakyReLU appended to ``buf``, conv5 with the fused skip epilogue) in one
    launch, for blocked bf16 buffers ``[N, H, 24, W <= 32, 8]``; bit-identical to five ``conv3x3_fwd`` calls."""
    a = dense_block_args(buf, w_packed, bias_packed, lrelu_slope, store_growth, res1, res1_scale, res2, res2_scale,
                         out_act, out_f32, mask, mask_unit_off, mask_slope, res1_weight, res2_weight, res_act)
    with _on(buf.device):
        launch_dense_block(a)


@dense_block_fwd.register_fake
def _(buf, w_packed, bias_packed, lrelu_slope, store_growth, res1, res1_scale, res2, res2_scale, out_act, out_f32,
      mask=None, mask_unit_off=None, mask_slope=1.0, res1_weight=1.0, res2_weight=1.0, res_act=None):
    return None


# ------------------------------------------------------------------------------------
@torch.library.custom_op("sr355::pack_weight_transposed", mutates_args=("w_packed",))
def pack_weight_transposed(w: Tensor, ci_lo: int, n_out: int, scale: float, w_packed: Tensor, packed_units: int,
                           unit_off: int, dtype: int) -> None:
    """Write the data-gradient (transposed, tap-mirrored) form of forward weight ``w`` into input units
    [unit_off, ...) of ``w_packed``; see sr355_pack_weight_transposed in include/sr355.h."""
    _need_cuda(w, w_packed)
    with _on(w.device):
        check(_lib.lib().sr355_pack_weight_transposed(w.data_ptr(), int(w.shape[0]), int(w.shape[1]), ci_lo, n_out,
                                                      dtype, float(scale), w_packed.data_ptr(), packed_units,
                                                      unit_off, _nbytes(w_packed), _stream()),
              "sr355_pack_weight_transposed")


@pack_weight_transposed.register_fake
def _(w, ci_lo, n_out, scale, w_packed, packed_units, unit_off, dtype):
    return None


def pack_weights_batch(weights: Sequence[Tensor], biases: Sequence[Tensor], dtype: int) -> List[Tuple[Tensor, Tensor]]:
    """``pack_weight`` of several convs in one launch (sr355_pack_weights): the packed weights are slices of one byte buffer,
    the padded biases slices of one fp32 buffer; same bytes as the single calls."""
    _need_cuda(*weights, *biases)
    L = _lib.lib()
    n = len(weights)
    sizes, pads = [], []
    for w in weights:
        if w.dtype != torch.float32 or w.dim() != 4 or w.shape[2:] != (3, 3) or not w.is_contiguous():
            raise RuntimeError("pack_weights_batch expects contiguous fp32 [cout, cin, 3, 3] tensors")
        nb = L.sr355_packed_weight_bytes(int(w.shape[0]), int(w.shape[1]), dtype, 0)
        if nb == 0:
            raise RuntimeError("pack_weights_batch: unsupported shape")
        sizes.append((nb + 255) // 256 * 256)
        pads.append((int(w.shape[0]) + 31) // 32 * 32)
    dev = weights[0].device
    blob = torch.empty(sum(sizes), dtype=torch.uint8, device=dev)
    bias_blob = torch.empty(sum(pads), dtype=torch.float32, device=dev)
    jobs = (_lib.PackJob * n)()
    out, at, bat = [], 0, 0
    for i, (w, b) in enumerate(zip(weights, biases)):
        wp, bp = blob[at:at + sizes[i]], bias_blob[bat:bat + pads[i]]
        j = jobs[i]
        j.w_dev, j.bias_dev, j.wp_dev, j.wp_bytes, j.bp_dev = w.data_ptr(), b.data_ptr(), wp.data_ptr(), sizes[i], bp.data_ptr()
        j.cout, j.cin, j.split = int(w.shape[0]), int(w.shape[1]), 0
        out.append((wp, bp))
        at += sizes[i]
        bat += pads[i]
    with _on(dev):
        check(L.sr355_pack_weights(jobs, n, dtype, _stream()), "sr355_pack_weights")
    return out


def pack_weights_transposed_batch(weights: Sequence[Tensor], dtype: int) -> List[Tuple[Tensor, Tensor]]:
    """The whole-conv data-gradient packs (``pack_weight_transposed(w, 0, cin, 1.0, ...)``) of several convs in one launch
    (sr355_pack_weights_transposed): slices of one zero-filled buffer, each with an all-zero bias of its output width."""
    _need_cuda(*weights)
    L = _lib.lib()
    n = len(weights)
    cpu = UNIT_CHANNELS[dtype]
    geo = []
    for w in weights:
        cout, cin = int(w.shape[0]), int(w.shape[1])
        n_out = (cin + 3) // 4 * 4                         # blocked outputs come in channel quads
        units = ((cout + cpu - 1) // cpu + 1) // 2 * 2
        pad = (n_out + 31) // 32 * 32
        geo.append((cout, cin, units, pad, (units * 9 * pad * 16 + 255) // 256 * 256))
    dev = weights[0].device
    blob = torch.zeros(sum(g[4] for g in geo), dtype=torch.uint8, device=dev)
    zero_bias = torch.zeros(max(g[3] for g in geo), dtype=torch.float32, device=dev)
    jobs = (_lib.PackTJob * n)()
    out, at = [], 0
    for i, (w, (cout, cin, units, pad, nb)) in enumerate(zip(weights, geo)):
        wt = blob[at:at + nb]
        j = jobs[i]
        j.w_dev, j.wp_dev, j.wp_bytes = w.data_ptr(), wt.data_ptr(), nb
        j.cout, j.cin, j.ci_lo, j.n_out, j.packed_units, j.unit_off, j.scale = cout, cin, 0, cin, units, 0, 1.0
        out.append((wt, zero_bias[:pad]))
        at += nb
    with _on(dev):
        check(L.sr355_pack_weights_transposed(jobs, n, dtype, _stream()), "sr355_pack_weights_transposed")
    return out


@torch.library.custom_op("sr355::pool2x2_sum", mutates_args=("out_f32", "out_act"))
def pool2x2_sum(x: Tensor, mask: Optional[Tensor], mask_unit_off: int, mask_slope: float,
                out_f32: Optional[Tensor], out_act: Optional[Tensor], out_unit_off: int) -> None:
    """Adjoint of nearest x2 upsampling on a blocked tensor [N,2H,U,2W,c], optionally sign-gated."""
    _need_cuda(x, mask, out_f32, out_act)
    N, H2, U, W2, _c = x.shape
    with _on(x.device):
        check(_lib.lib().sr355_pool2x2_sum(
            x.data_ptr(), _nbytes(x), N, H2 // 2, W2 // 2, U, _CODE[x.dtype],
            _ptr(mask), _nbytes(mask), int(mask.shape[2]) if mask is not None else 0, mask_unit_off, float(mask_slope),
            _ptr(out_f32), _nbytes(out_f32), _ptr(out_act), _nbytes(out_act),
            int(out_act.shape[2]) if out_act is not None else 0, out_unit_off, _stream()), "sr355_pool2x2_sum")


@pool2x2_sum.register_fake
def _(x, mask, mask_unit_off, mask_slope, out_f32, out_act, out_unit_off):
    return None


@torch.library.custom_op("sr355::conv3x3_wgrad", mutates_args=("dw", "db"))
def conv3x3_wgrad(in_buf: Tensor, cin: int, upsample: int, dy: Tensor, dy_unit_off: int, cout: int, scale: float,
                  accumulate: int, dw: Tensor, db: Optional[Tensor]) -> None:
    """Weight / bias gradient of a 3x3 conv from its blocked bf16 input and output gradient."""
    _need_cuda(in_buf, dy, dw, db)
    if in_buf.dtype != torch.bfloat16 or dy.dtype != torch.bfloat16:
        raise RuntimeError("conv3x3_wgrad runs on bf16 activations")
    a = WgradArgs()
    N, H, _U, W, _c = dy.shape
    a.N, a.H, a.W, a.upsample = int(N), int(H), int(W), int(upsample)
    a.in_dev, a.in_bytes, a.in_units, a.cin = in_buf.data_ptr(), _nbytes(in_buf), int(in_buf.shape[2]), int(cin)
    a.dy_dev, a.dy_bytes, a.dy_units, a.dy_unit_off = dy.data_ptr(), _nbytes(dy), int(dy.shape[2]), int(dy_unit_off)
    a.cout, a.scale, a.accumulate = int(cout), float(scale), int(accumulate)
    a.dw_dev, a.dw_bytes, a.db_dev = dw.data_ptr(), _nbytes(dw), _ptr(db)
    L = _lib.lib()
    need = L.sr355_wgrad_workspace_bytes(cout, cin, N, H, W)
    if need == 0:
        raise RuntimeError("conv3x3_wgrad: unsupported shape")
    ws = torch.empty(need, dtype=torch.uint8, device=dy.device)
    a.workspace_dev, a.workspace_bytes = ws.data_ptr(), need
    with _on(dy.device):
        check(L.sr355_conv3x3_wgrad(C.byref(a), _stream()), "sr355_conv3x3_wgrad")


@conv3x3_wgrad.register_fake
def _(in_buf, cin, upsample, dy, dy_unit_off, cout, scale, accumulate, dw, db):
    return None


@torch.library.custom_op("sr355::conv3x3_wgrad_f32", mutates_args=("dw", "db"))
def conv3x3_wgrad_f32(in_buf: Tensor, cin: int, upsample: int, dy: Tensor, dy_unit_off: int, cout: int, scale: float,
                      accumulate: int, dw: Tensor, db: Optional[Tensor]) -> None:
    """Weight / bias gradient of a 3x3 conv on the exact-fp32 path: blocked fp32 input and output gradient
    ``[N, H, units, W, 4]`` (``dy_unit_off`` in 4-channel units), any number of output channels."""
    _need_cuda(in_buf, dy, dw, db)
    if in_buf.dtype != torch.float32 or dy.dtype != torch.float32 or in_buf.dim() != 5 or dy.dim() != 5:
        raise RuntimeError("conv3x3_wgrad_f32 runs on blocked fp32 tensors [N, H, units, W, 4]")
    a = WgradArgs()
    N, H, _U, W, _c = dy.shape
    a.N, a.H, a.W, a.upsample = int(N), int(H), int(W), int(upsample)
    a.in_dev, a.in_bytes, a.in_units, a.cin = in_buf.data_ptr(), _nbytes(in_buf), int(in_buf.shape[2]), int(cin)
    a.dy_dev, a.dy_bytes, a.dy_units, a.dy_unit_off = dy.data_ptr(), _nbytes(dy), int(dy.shape[2]), int(dy_unit_off)
    a.cout, a.scale, a.accumulate = int(cout), float(scale), int(accumulate)
    a.dw_dev, a.dw_bytes, a.db_dev = dw.data_ptr(), _nbytes(dw), _ptr(db)
    L = _lib.lib()
    need = L.sr355_wgrad_f32_workspace_bytes(cout, cin, N, H, W)
    if need == 0:
        raise RuntimeError("conv3x3_wgrad_f32: unsupported shape")
    ws = torch.empty(need, dtype=torch.uint8, device=dy.device)
    a.workspace_dev, a.workspace_bytes = ws.data_ptr(), need
    with _on(dy.device):
        check(L.sr355_conv3x3_wgrad_f32(C.byref(a), _stream()), "sr355_conv3x3_wgrad_f32")


@conv3x3_wgrad_f32.register_fake
def _(in_buf, cin, upsample, dy, dy_unit_off, cout, scale, accumulate, dw, db):
    return None


def dense_wgrad_args(act: Tensor, stack: Tensor, nf: int, gc: int, growth_scale: float, conv5_scale: float,
                     dw: List[Tensor], db: List[Tensor], workspace: Tensor) -> DenseWgradArgs:
    """The sr355_dense_wgrad_args record of one batched dense-block weight-gradient launch."""
    _need_cuda(act, stack, workspace, *dw, *db)
    if act.dtype != torch.bfloat16 or stack.dtype != torch.bfloat16 or act.shape != stack.shape or act.dim() != 5:
        raise RuntimeError("dense_block_wgrad: act and stack must be blocked bf16 buffers of one geometry")
    if len(dw) != 5 or len(db) != 5 or any(t.dtype != torch.float32 or not t.is_contiguous() for t in (*dw, *db)):
        raise RuntimeError("dense_block_wgrad: five contiguous fp32 dw and db tensors")
    a = DenseWgradArgs()
    a.N, a.H, a.W, a.units = int(act.shape[0]), int(act.shape[1]), int(act.shape[3]), int(act.shape[2])
    a.act_dev, a.act_bytes, a.stack_dev, a.stack_bytes = act.data_ptr(), _nbytes(act), stack.data_ptr(), _nbytes(stack)
    a.num_features, a.growth_channels = int(nf), int(gc)
    a.growth_scale, a.conv5_scale = float(growth_scale), float(conv5_scale)
    for k in range(5):
        cin, cout = nf + k * gc, (gc if k < 4 else nf)
        if tuple(dw[k].shape) != (cout, cin, 3, 3) or db[k].numel() != cout:
            raise RuntimeError(f"dense_block_wgrad: dw[{k}] / db[{k}] have the wrong shape")
        a.dw_dev[k], a.db_dev[k] = dw[k].data_ptr(), db[k].data_ptr()
    a.workspace_dev, a.workspace_bytes = workspace.data_ptr(), _nbytes(workspace)
    return a


@torch.library.custom_op("sr355::dense_block_wgrad", mutates_args=("dw", "db"))
def dense_block_wgrad(act: Tensor, stack: Tensor, nf: int, gc: int, growth_scale: float, conv5_scale: float,
                      dw: List[Tensor], db: List[Tensor]) -> None:
    """Weight / bias gradients of the five convs of a dense block from its forward buffer and its gradient stack
    ``[dOut | dY4 | dY3 | dY2 | dY1]``: one batched launch and one reduction, bit-identical to five ``conv3x3_wgrad``."""
    L = _lib.lib()
    need = L.sr355_dense_block_wgrad_workspace_bytes(int(nf), int(gc), int(act.shape[0]), int(act.shape[1]), int(act.shape[3]))
    if need == 0:
        raise RuntimeError("dense_block_wgrad: unsupported shape")
    ws = torch.empty(need, dtype=torch.uint8, device=act.device)
    a = dense_wgrad_args(act, stack, nf, gc, growth_scale, conv5_scale, dw, db, ws)
    with _on(act.device):
        check(L.sr355_dense_block_wgrad(C.byref(a), _stream()), "sr355_dense_block_wgrad")


@dense_block_wgrad.register_fake
def _(act, stack, nf, gc, growth_scale, conv5_scale, dw, db):
    return None


def conv3x3_wgrad_all(in_buf: Tensor, cin: int, dy: Tensor, cout: int, dw: Tensor, db: Tensor,
                      stride: int = 1) -> None:
    """Weight / bias gradient of a conv with any number of output channels: one ``sr355_conv3x3_wgrad`` call per 4096
    outputs (the library runs the 32-channel slices of a wide layer as the jobs of one batched launch)."""
    _need_cuda(in_buf, dy, dw, db)
    if in_buf.dtype != torch.bfloat16 or dy.dtype != torch.bfloat16:
        raise RuntimeError("conv3x3_wgrad runs on bf16 activations")
    if not (dw.is_contiguous() and db.is_contiguous()) or dw.shape[0] != cout or db.numel() != cout:
        raise RuntimeError("conv3x3_wgrad_all: dw [cout, cin, 3, 3] and db [cout] must be contiguous fp32 tensors")
    L = _lib.lib()
    N, H, _U, W, _c = (int(v) for v in (dy.shape if stride == 1 else in_buf.shape))   # stride 2: the input's size
    a = WgradArgs()
    a.N, a.H, a.W, a.upsample = N, H, W, 0
    a.dy_stride2 = 1 if stride == 2 else 0
    a.in_dev, a.in_bytes, a.in_units, a.cin = in_buf.data_ptr(), _nbytes(in_buf), int(in_buf.shape[2]), int(cin)
    a.dy_dev, a.dy_bytes, a.dy_units = dy.data_ptr(), _nbytes(dy), int(dy.shape[2])
    a.scale, a.accumulate = 1.0, 0
    step = min(cout, 4096)
    need = L.sr355_wgrad_workspace_bytes(step, cin, N, H, W)
    if need == 0:
        raise RuntimeError("conv3x3_wgrad: unsupported shape")
    ws = torch.empty(need, dtype=torch.uint8, device=dy.device)      # calls run in stream order: one workspace serves all
    a.workspace_dev, a.workspace_bytes = ws.data_ptr(), need
    dw_ptr, db_ptr, row = dw.data_ptr(), db.data_ptr(), int(cin) * 9 * 4
    with _on(dy.device):
        stream = _stream()
        for c0 in range(0, cout, step):
            cb = min(step, cout - c0)
            a.dy_unit_off, a.cout = c0 // 8, cb
            a.dw_dev, a.dw_bytes, a.db_dev = dw_ptr + c0 * row, cb * row, db_ptr + c0 * 4
            check(L.sr355_conv3x3_wgrad(C.byref(a), stream), "sr355_conv3x3_wgrad")


# ------------------------------------------------------------------------------------
# Layers around the convs of the discriminator / VGG19 and the loss reductions (csrc/aux_layers.hip)
@torch.library.custom_op("sr355::bn_lrelu_fwd", mutates_args=("running_mean", "running_var", "num_batches_tracked",
                                                                "save_mean", "save_rstd", "out_act"))
def bn_lrelu_fwd(x_f32: Tensor, gamma: Tensor, beta: Tensor, eps: float, slope: float, training: bool, momentum: float,
                 running_mean: Optional[Tensor], running_var: Optional[Tensor], num_batches_tracked: Optional[Tensor],
                 save_mean: Tensor, save_rstd: Tensor, out_act: Tensor, out_unit_off: int = 0, groups: int = 1) -> None:
    """BatchNorm2d (batch statistics when ``training``) + LeakyReLU of an fp32 conv output ``[N, H, C/4, W, 4]`` into
    blocked bf16 ``out_act``; updates the running buffers like nn.BatchNorm2d.  ``groups``: the batch stands for that
    many calls of the layer on its consecutive sub-batches (statistics ``[groups, C]``, the running buffers updated once
    per group in order), bit-identical to the separate calls."""
    _need_cuda(x_f32, gamma, beta, running_mean, running_var, num_batches_tracked, save_mean, save_rstd, out_act)
    N, H, Q, W, _c = (int(v) for v in x_f32.shape)
    groups = int(groups)
    if groups < 1 or N % groups or save_mean.numel() != groups * 4 * Q or save_rstd.numel() != groups * 4 * Q:
        raise RuntimeError("bn_lrelu_fwd: the groups must divide the batch and the statistics be [groups, C]")
    if not (x_f32.is_contiguous() and save_mean.is_contiguous() and save_rstd.is_contiguous()):
        raise RuntimeError("bn_lrelu_fwd: contiguous tensors expected")
    L = _lib.lib()
    need = groups * L.sr355_bn_workspace_bytes(N // groups, H, W, 4 * Q)
    if need == 0:
        raise RuntimeError("bn_lrelu_fwd: channels must be a positive multiple of 8")
    ws = torch.empty(need, dtype=torch.uint8, device=x_f32.device)
    with _on(x_f32.device):
        check(L.sr355_bn_lrelu_fwd_groups(x_f32.data_ptr(), _nbytes(x_f32), N, H, W, 4 * Q, gamma.data_ptr(),
                                          beta.data_ptr(), float(eps), float(slope), int(bool(training)), float(momentum),
                                          _ptr(running_mean), _ptr(running_var), _ptr(num_batches_tracked),
                                          save_mean.data_ptr(), save_rstd.data_ptr(), out_act.data_ptr(), _nbytes(out_act),
                                          int(out_act.shape[2]), int(out_unit_off), ws.data_ptr(), need, groups, _stream()),
              "sr355_bn_lrelu_fwd_groups")


@bn_lrelu_fwd.register_fake
def _(x_f32, gamma, beta, eps, slope, training, momentum, running_mean, running_var, num_batches_tracked, save_mean,
      save_rstd, out_act, out_unit_off=0, groups=1):
    return None


@torch.library.custom_op("sr355::bn_lrelu_bwd", mutates_args=("dx_act", "dgamma", "dbeta"))
def bn_lrelu_bwd(x_f32: Tensor, dy_act: Tensor, gamma: Tensor, beta: Tensor, save_mean: Tensor, save_rstd: Tensor,
                 slope: float, training: bool, dx_act: Tensor, dgamma: Tensor, dbeta: Tensor, dy_unit_off: int = 0,
                 dx_unit_off: int = 0, groups: int = 1) -> None:
    """Backward of ``bn_lrelu_fwd``: the gradient of x as blocked bf16, and dgamma / dbeta (``[groups, C]``: one row per
    sub-batch, as the separate calls would return them).  ``dy_act`` is the blocked bf16 gradient of the output or an fp32
    tensor ``[N, H, C/4, W, 4]`` (a data-gradient conv's ``out_f32``)."""
    _need_cuda(x_f32, dy_act, gamma, beta, save_mean, save_rstd, dx_act, dgamma, dbeta)
    N, H, Q, W, _c = (int(v) for v in x_f32.shape)
    dy_f32 = dy_act.dtype == torch.float32
    if dy_f32 and (tuple(dy_act.shape) != tuple(x_f32.shape) or dy_unit_off != 0 or not dy_act.is_contiguous()):
        raise RuntimeError("bn_lrelu_bwd: an fp32 gradient must be a contiguous tensor of x's shape [N, H, C/4, W, 4]")
    groups = int(groups)
    n_stat = groups * 4 * Q
    if groups < 1 or N % groups or any(t.numel() != n_stat or not t.is_contiguous()
                                       for t in (save_mean, save_rstd, dgamma, dbeta)):
        raise RuntimeError("bn_lrelu_bwd: the groups must divide the batch and statistics / sums be contiguous [groups, C]")
    L = _lib.lib()
    need = groups * L.sr355_bn_workspace_bytes(N // groups, H, W, 4 * Q)
    ws = torch.empty(need, dtype=torch.uint8, device=x_f32.device)
    with _on(x_f32.device):
        check(L.sr355_bn_lrelu_bwd_groups(x_f32.data_ptr(), _nbytes(x_f32), dy_act.data_ptr(), _nbytes(dy_act),
                                          0 if dy_f32 else int(dy_act.shape[2]), int(dy_unit_off), N, H, W, 4 * Q,
                                          gamma.data_ptr(), beta.data_ptr(), save_mean.data_ptr(), save_rstd.data_ptr(),
                                          float(slope), int(bool(training)), dx_act.data_ptr(), _nbytes(dx_act),
                                          int(dx_act.shape[2]), int(dx_unit_off), dgamma.data_ptr(), dbeta.data_ptr(),
                                          ws.data_ptr(), need, groups, _stream()),
              "sr355_bn_lrelu_bwd_groups")


@bn_lrelu_bwd.register_fake
def _(x_f32, dy_act, gamma, beta, save_mean, save_rstd, slope, training, dx_act, dgamma, dbeta, dy_unit_off=0,
      dx_unit_off=0, groups=1):
    return None


@torch.library.custom_op("sr355::pixel_shuffle2", mutates_args=())
def pixel_shuffle2(x: Tensor, inverse: bool = False) -> Tensor:
    """``nn.PixelShuffle(2)`` of a blocked tensor ``[N, H, 4U, W, c]`` -> ``[N, 2H, U, 2W, c]`` (bf16 or fp32 units);
    ``inverse`` maps ``[N, 2H, U, 2W, c]`` back to ``[N, H, 4U, W, c]`` (its gradient)."""
    _need_cuda(x)
    if x.dim() != 5 or x.dtype not in _CODE or x.shape[4] != UNIT_CHANNELS[_CODE[x.dtype]]:
        raise RuntimeError("pixel_shuffle2 runs on blocked bf16 / fp32 tensors [N, H, units, W, c]")
    N, Hx, Ux, Wx, c = (int(v) for v in x.shape)
    if inverse:
        if Hx % 2 or Wx % 2:
            raise RuntimeError("pixel_shuffle2 (inverse): height and width must be even")
        H, W, Uo = Hx // 2, Wx // 2, Ux
        out = torch.empty((N, H, 4 * Uo, W, c), dtype=x.dtype, device=x.device)
    else:
        if Ux % 4:
            raise RuntimeError("pixel_shuffle2: the unit count must be a multiple of 4")
        H, W, Uo = Hx, Wx, Ux // 4
        out = torch.empty((N, 2 * H, Uo, 2 * W, c), dtype=x.dtype, device=x.device)
    with _on(x.device):
        check(_lib.lib().sr355_pixel_shuffle2(x.data_ptr(), _nbytes(x), N, H, W, Uo * c, _CODE[x.dtype], int(inverse),
                                              out.data_ptr(), _nbytes(out), _stream()), "sr355_pixel_shuffle2")
    return out


@pixel_shuffle2.register_fake
def _(x, inverse=False):
    N, Hx, Ux, Wx, c = x.shape
    return x.new_empty((N, Hx // 2, 4 * Ux, Wx // 2, c) if inverse else (N, 2 * Hx, Ux // 4, 2 * Wx, c))


@torch.library.custom_op("sr355::maxpool2x2_fwd", mutates_args=())
def maxpool2x2_fwd(x: Tensor) -> Tensor:
    """nn.MaxPool2d(2, 2) on a blocked bf16 tensor ``[N, H, U, W, 8]`` -> ``[N, H//2, U, W//2, 8]``."""
    _need_cuda(x)
    if x.dtype != torch.bfloat16 or x.dim() != 5:
        raise RuntimeError("maxpool2x2 runs on blocked bf16 tensors")
    N, H, U, W, _c = (int(v) for v in x.shape)
    out = torch.empty((N, H // 2, U, W // 2, 8), dtype=torch.bfloat16, device=x.device)
    with _on(x.device):
        check(_lib.lib().sr355_maxpool2x2_fwd(x.data_ptr(), _nbytes(x), N, H, W, U, out.data_ptr(), _nbytes(out),
                                              _stream()), "sr355_maxpool2x2_fwd")
    return out


@maxpool2x2_fwd.register_fake
def _(x):
    N, H, U, W, c = x.shape
    return x.new_empty((N, H // 2, U, W // 2, c))


@torch.library.custom_op("sr355::maxpool2x2_bwd", mutates_args=())
def maxpool2x2_bwd(x: Tensor, dy: Tensor) -> Tensor:
    """Gradient of ``maxpool2x2_fwd`` with respect to its input ``x`` (first maximum of a window takes it all)."""
    _need_cuda(x, dy)
    N, H, U, W, _c = (int(v) for v in x.shape)
    dx = torch.empty_like(x)
    with _on(x.device):
        check(_lib.lib().sr355_maxpool2x2_bwd(x.data_ptr(), _nbytes(x), dy.data_ptr(), _nbytes(dy), N, H, W, U,
                                              dx.data_ptr(), _nbytes(dx), _stream()), "sr355_maxpool2x2_bwd")
    return dx


@maxpool2x2_bwd.register_fake
def _(x, dy):
    return torch.empty_like(x)


@torch.library.custom_op("sr355::loss_fwd", mutates_args=())
def loss_fwd(a: Tensor, b: Optional[Tensor], b_const: float, mode: int, want_grad: bool) -> Tuple[Tensor, Tensor]:
    """mean |a - b| (mode 0) or mean (a - b)^2 (mode 1) of fp32 tensors (``b`` None: the constant ``b_const``) and, when
    ``want_grad``, d(loss)/da from the same pass (else an empty tensor)."""
    _need_cuda(a, b)
    if a.dtype != torch.float32 or (b is not None and (b.dtype != torch.float32 or b.shape != a.shape)):
        raise RuntimeError("loss_fwd: fp32 tensors of one shape")
    L = _lib.lib()
    n = a.numel()
    loss = torch.empty((), dtype=torch.float32, device=a.device)
    grad = torch.empty_like(a) if want_grad else torch.empty(0, dtype=torch.float32, device=a.device)
    need = L.sr355_loss_workspace_bytes(n)
    ws = torch.empty(max(need, 4), dtype=torch.uint8, device=a.device)
    with _on(a.device):
        check(L.sr355_loss_fwd(a.data_ptr(), _ptr(b), float(b_const), n, int(mode), loss.data_ptr(),
                               grad.data_ptr() if want_grad else None, ws.data_ptr(), need, _stream()), "sr355_loss_fwd")
    return loss, grad


@loss_fwd.register_fake
def _(a, b, b_const, mode, want_grad):
    return a.new_empty(()), (torch.empty_like(a) if want_grad else a.new_empty(0))


# ------------------------------------------------------------------------------------
def make_cfg(cfg: List[int]) -> GeneratorCfg:
    """cfg = [in_channels, out_channels, num_features, num_blocks, growth_channels, dtype, hr_split] and optionally
    trunk_fp32 (default 1: fp32 carriers for every skip)."""
    c = GeneratorCfg()
    (c.in_channels, c.out_channels, c.num_features, c.num_blocks, c.growth_channels, c.dtype,
     c.hr_split) = [int(v) for v in cfg[:7]]
    c.trunk_fp32 = int(cfg[7]) if len(cfg) > 7 else 1
    return c


def generator_packed_bytes(cfg: List[int]) -> int:
    c = make_cfg(cfg)
    n = _lib.lib().sr355_generator_packed_bytes(C.byref(c))
    if n == 0:
        check(-1, "sr355_generator_packed_bytes")
    return n


def generator_workspace_bytes(cfg: List[int], N: int, H: int, W: int) -> int:
    c = make_cfg(cfg)
    n = _lib.lib().sr355_generator_workspace_bytes(C.byref(c), N, H, W)
    if n == 0:
        check(-1, "sr355_generator_workspace_bytes")
    return n


def generator_pack_now(weights: List[Tensor], biases: List[Tensor], cfg: List[int]) -> Tensor:
    """What ``sr355::generator_pack`` runs, callable without the dispatcher: the op reads the parameters' memory
    and nothing else of them, so a module that repacks 702 tensors every training step calls this with its
    Parameters as they are (no detached aliases, no per-tensor dispatch bookkeeping)."""
    _need_cuda(*weights, *biases)
    c = make_cfg(cfg)
    L = _lib.lib()
    n = len(weights)
    nbytes = generator_packed_bytes(cfg)
    blob = torch.empty(nbytes, dtype=torch.uint8, device=weights[0].device)
    wp = (C.c_void_p * n)(*[w.data_ptr() for w in weights])
    bp = (C.c_void_p * n)(*[b.data_ptr() for b in biases])
    for w in weights + biases:
        if w.dtype != torch.float32:
            raise RuntimeError("generator_pack expects fp32 parameters")
    with _on(blob.device):
        check(L.sr355_generator_pack(C.byref(c), wp, bp, n, blob.data_ptr(), nbytes, _stream()),
              "sr355_generator_pack")
    return blob


@torch.library.custom_op("sr355::generator_pack", mutates_args=())
def generator_pack(weights: List[Tensor], biases: List[Tensor], cfg: List[int]) -> Tensor:
    """Pack all conv parameters (state_dict order) into one device blob."""
    return generator_pack_now(weights, biases, cfg)


@generator_pack.register_fake
def _(weights, biases, cfg):
    return weights[0].new_empty(generator_packed_bytes(cfg), dtype=torch.uint8)


@torch.library.custom_op("sr355::generator_forward", mutates_args=("workspace",))
def generator_forward(x: Tensor, packed: Tensor, workspace: Tensor, cfg: List[int]) -> Tensor:
    """GeneratorRRDB.forward: [N,in,H,W] fp32 -> [N,out,4H,4W] fp32, unclamped."""
    _need_cuda(x, packed, workspace)
    if x.dtype != torch.float32 or x.dim() != 4:
        raise RuntimeError("generator_forward expects an fp32 NCHW tensor")
    c = make_cfg(cfg)
    N, Cin, H, W = (int(v) for v in x.shape)
    if Cin != c.in_channels:
        raise RuntimeError(f"generator_forward: input has {Cin} channels, model expects {c.in_channels}")
    out = torch.empty((N, c.out_channels, 4 * H, 4 * W), dtype=torch.float32, device=x.device)
    with _on(x.device):
        check(_lib.lib().sr355_generator_forward(C.byref(c), packed.data_ptr(), _nbytes(packed), x.data_ptr(),
                                                 out.data_ptr(), N, H, W, workspace.data_ptr(),
                                                 _nbytes(workspace), _stream()), "sr355_generator_forward")
    return out


@generator_forward.register_fake
def _(x, packed, workspace, cfg):
    N, _c, H, W = x.shape
    return x.new_empty((N, cfg[1], 4 * H, 4 * W), dtype=torch.float32)


@torch.library.custom_op("sr355::generator_forward_u8", mutates_args=("workspace",))
def generator_forward_u8(image: Tensor, packed: Tensor, workspace: Tensor, cfg: List[int]) -> Tensor:
    """The inference loop body of /root/reference/esrgan/test.py:68-74 between image buffers: uint8 [N,H,W,in]
    (HWC, as PIL / numpy hold an image) read as byte / 255 -> uint8 [N,4H,4W,out] = trunc(255 * clamp(y, 0, 1))."""
    _need_cuda(image, packed, workspace)
    if image.dtype != torch.uint8 or image.dim() != 4 or not image.is_contiguous():
        raise RuntimeError("generator_forward_u8 expects a contiguous uint8 NHWC tensor")
    c = make_cfg(cfg)
    N, H, W, Cin = (int(v) for v in image.shape)
    if Cin != c.in_channels:
        raise RuntimeError(f"generator_forward_u8: image has {Cin} channels, model expects {c.in_channels}")
    out = torch.empty((N, 4 * H, 4 * W, c.out_channels), dtype=torch.uint8, device=image.device)
    with _on(image.device):
        check(_lib.lib().sr355_generator_forward_u8(C.byref(c), packed.data_ptr(), _nbytes(packed), image.data_ptr(),
                                                    out.data_ptr(), N, H, W, workspace.data_ptr(),
                                                    _nbytes(workspace), _stream()), "sr355_generator_forward_u8")
    return out


@generator_forward_u8.register_fake
def _(image, packed, workspace, cfg):
    N, H, W, _c = image.shape
    return image.new_empty((N, 4 * H, 4 * W, cfg[1]), dtype=torch.uint8)


# ------------------------------------------------------------------------------------
# training: forward that keeps activations + backward
def _size(fn_name: str, cfg: List[int], *shape: int) -> int:
    c = make_cfg(cfg)
    n = getattr(_lib.lib(), fn_name)(C.byref(c), *shape)
    if n == 0:
        check(-1, fn_name)
    return n


def generator_saved_bytes(cfg: List[int], N: int, H: int, W: int) -> int:
    return _size("sr355_generator_saved_bytes", cfg, N, H, W)


def generator_train_workspace_bytes(cfg: List[int], N: int, H: int, W: int) -> int:
    return _size("sr355_generator_train_workspace_bytes", cfg, N, H, W)


def generator_bwd_workspace_bytes(cfg: List[int], N: int, H: int, W: int) -> int:
    return _size("sr355_generator_bwd_workspace_bytes", cfg, N, H, W)


def generator_grads_floats(cfg: List[int]) -> int:
    return _size("sr355_generator_grads_floats", cfg)


@torch.library.custom_op("sr355::generator_forward_train", mutates_args=("saved", "workspace"))
def generator_forward_train(x: Tensor, packed: Tensor, saved: Tensor, workspace: Tensor, cfg: List[int]) -> Tensor:
    """GeneratorRRDB.forward that also fills ``saved`` with the activations the backward pass reads."""
    _need_cuda(x, packed, saved, workspace)
    c = make_cfg(cfg)
    N, _cin, H, W = (int(v) for v in x.shape)
    out = torch.empty((N, c.out_channels, 4 * H, 4 * W), dtype=torch.float32, device=x.device)
    with _on(x.device):
        check(_lib.lib().sr355_generator_forward_train(C.byref(c), packed.data_ptr(), _nbytes(packed), x.data_ptr(),
                                                       out.data_ptr(), N, H, W, saved.data_ptr(), _nbytes(saved),
                                                       workspace.data_ptr(), _nbytes(workspace), _stream()),
              "sr355_generator_forward_train")
    return out


@generator_forward_train.register_fake
def _(x, packed, saved, workspace, cfg):
    N, _c, H, W = x.shape
    return x.new_empty((N, cfg[1], 4 * H, 4 * W), dtype=torch.float32)


def generator_pack_bwd_now(weights: List[Tensor], cfg: List[int]) -> Tensor:
    """What ``sr355::generator_pack_bwd`` runs, callable without the dispatcher (see ``generator_pack_now``)."""
    _need_cuda(*weights)
    c = make_cfg(cfg)
    L = _lib.lib()
    nbytes = L.sr355_generator_packed_bwd_bytes(C.byref(c))
    if nbytes == 0:
        check(-1, "sr355_generator_packed_bwd_bytes")
    blob = torch.empty(nbytes, dtype=torch.uint8, device=weights[0].device)
    n = len(weights)
    wp = (C.c_void_p * n)(*[w.data_ptr() for w in weights])
    with _on(blob.device):
        check(L.sr355_generator_pack_bwd(C.byref(c), wp, n, blob.data_ptr(), nbytes, _stream()),
              "sr355_generator_pack_bwd")
    return blob


@torch.library.custom_op("sr355::generator_pack_bwd", mutates_args=())
def generator_pack_bwd(weights: List[Tensor], cfg: List[int]) -> Tensor:
    """Transposed (data-gradient) packing of every conv weight (state_dict order)."""
    return generator_pack_bwd_now(weights, cfg)


@generator_pack_bwd.register_fake
def _(weights, cfg):
    c = make_cfg(cfg)
    return weights[0].new_empty(_lib.lib().sr355_generator_packed_bwd_bytes(C.byref(c)), dtype=torch.uint8)


@torch.library.custom_op("sr355::generator_backward", mutates_args=("workspace",))
def generator_backward(grad_out: Tensor, packed_bwd: Tensor, saved: Tensor, workspace: Tensor, cfg: List[int],
                       lr_height: int, lr_width: int, need_input_grad: bool) -> Tuple[Tensor, Tensor]:
    """Returns (flat fp32 parameter gradients: per conv weight then bias, gradient of the input or an
    empty tensor)."""
    _need_cuda(grad_out, packed_bwd, saved, workspace)
    c = make_cfg(cfg)
    N = int(grad_out.shape[0])
    H, W = int(lr_height), int(lr_width)
    L = _lib.lib()
    nfl = L.sr355_generator_grads_floats(C.byref(c))
    grads = torch.empty(nfl, dtype=torch.float32, device=grad_out.device)
    gin = (torch.empty((N, c.in_channels, H, W), dtype=torch.float32, device=grad_out.device)
           if need_input_grad else torch.empty(0, dtype=torch.float32, device=grad_out.device))
    with _on(grad_out.device):
        check(L.sr355_generator_backward(C.byref(c), packed_bwd.data_ptr(), _nbytes(packed_bwd), saved.data_ptr(),
                                         _nbytes(saved), grad_out.data_ptr(), grads.data_ptr(), nfl,
                                         gin.data_ptr() if need_input_grad else None, N, H, W,
                                         workspace.data_ptr(), _nbytes(workspace), _stream()),
              "sr355_generator_backward")
    return grads, gin


@generator_backward.register_fake
def _(grad_out, packed_bwd, saved, workspace, cfg, lr_height, lr_width, need_input_grad):
    c = make_cfg(cfg)
    nfl = _lib.lib().sr355_generator_grads_floats(C.byref(c))
    N = grad_out.shape[0]
    gin = grad_out.new_empty((N, cfg[0], lr_height, lr_width) if need_input_grad else (0,))
    return grad_out.new_empty(nfl), gin


@torch.library.custom_op("sr355::generator_backward_range", mutates_args=("grads", "grad_in", "workspace"))
def generator_backward_range(grad_out: Tensor, packed_bwd: Tensor, saved: Tensor, workspace: Tensor, grads: Tensor,
                             grad_in: Optional[Tensor], cfg: List[int], lr_height: int, lr_width: int,
                             phase_begin: int, phase_end: int) -> None:
    """Phases [phase_begin, phase_end) of the backward pass (see sr355_generator_backward_range); each
    phase writes the gradients of its own convs into the flat ``grads`` buffer."""
    _need_cuda(grad_out, packed_bwd, saved, workspace, grads, grad_in)
    c = make_cfg(cfg)
    with _on(grad_out.device):
        check(_lib.lib().sr355_generator_backward_range(
            C.byref(c), packed_bwd.data_ptr(), _nbytes(packed_bwd), saved.data_ptr(), _nbytes(saved),
            grad_out.data_ptr(), grads.data_ptr(), grads.numel(), _ptr(grad_in), int(grad_out.shape[0]),
            int(lr_height), int(lr_width), workspace.data_ptr(), _nbytes(workspace), int(phase_begin), int(phase_end),
            _stream()), "sr355_generator_backward_range")


@generator_backward_range.register_fake
def _(grad_out, packed_bwd, saved, workspace, grads, grad_in, cfg, lr_height, lr_width, phase_begin, phase_end):
    return None


# ------------------------------------------------------------------------------------
# optimizer tail
@torch.library.custom_op("sr355::adam_step", mutates_args=("param", "exp_avg", "exp_avg_sq"))
def adam_step(param: Tensor, grad: Tensor, exp_avg: Tensor, exp_avg_sq: Tensor, lr: float, beta1: float,
              beta2: float, eps: float, weight_decay: float, step: int) -> None:
    """One torch.optim.Adam update (esrgan/train.py:117-122) over flat fp32 buffers, one launch."""
    _need_cuda(param, grad, exp_avg, exp_avg_sq)
    n = param.numel()
    for t in (param, grad, exp_avg, exp_avg_sq):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != n:
            raise RuntimeError("adam_step expects four contiguous fp32 tensors of one length")
    with _on(param.device):
        check(_lib.lib().sr355_adam_step(param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(),
                                         exp_avg_sq.data_ptr(), n, lr, beta1, beta2, eps, weight_decay, step,
                                         _stream()), "sr355_adam_step")


@adam_step.register_fake
def _(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step):
    return None


def now(op):
    """The Python body of a registered ``sr355::`` op, callable without the dispatcher.  Inside an
    ``autograd.Function`` nothing is recorded and the ops only read their tensors' memory, so the per-call
    bookkeeping of the dispatcher (schema walk, autograd keys, alias checks: several microseconds, paid
    hundreds of times per training step) buys nothing there.  Falls back to the op itself."""
    return getattr(op, "_init_fn", op)


def chain_status(wait: bool = True, device=None) -> bool:
    """Health of the most recent whole-generator forward that ran its body as ONE persistent launch on ``device``
    (default: the current one).  That launch needs all its workgroups resident; where they cannot be (a second
    process on the GPU) a wait gives up after two seconds and the forward's output is invalid.  Returns True when the
    launch completed soundly (or none ran), False when ``wait`` is False and it has not finished yet; raises
    ``Sr355Error`` once for a launch that gave up — the device then runs one launch per conv
    (``chain_reenable()`` undoes that).  ``wait=True`` blocks until the launch has completed: use it before trusting
    the output of a one-shot forward."""
    with _on(torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)):
        rc = _lib.lib().sr355_chain_status(1 if wait else 0)
    if rc == 1:
        return False
    check(rc, "sr355_chain_status")
    return True


def chain_reenable(device=None) -> None:
    """Let the body run as one persistent launch again on ``device`` after a launch that gave up."""
    with _on(torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)):
        _lib.lib().sr355_chain_reenable()


conv3x3_fwd_now = now(conv3x3_fwd)
bn_lrelu_fwd_now = now(bn_lrelu_fwd)
bn_lrelu_bwd_now = now(bn_lrelu_bwd)
maxpool2x2_fwd_now = now(maxpool2x2_fwd)
pixel_shuffle2_now = now(pixel_shuffle2)
maxpool2x2_bwd_now = now(maxpool2x2_bwd)
loss_fwd_now = now(loss_fwd)
conv3x3_wgrad_now = now(conv3x3_wgrad)
dense_block_wgrad_now = now(dense_block_wgrad)
conv3x3_wgrad_f32_now = now(conv3x3_wgrad_f32)
pack_weight_now = now(pack_weight)
pack_weight_transposed_now = now(pack_weight_transposed)
pool2x2_sum_now = now(pool2x2_sum)
to_blocked_now = now(to_blocked)
from_blocked_now = now(from_blocked)
