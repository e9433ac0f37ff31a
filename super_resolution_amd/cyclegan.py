"""The CycleGAN super-resolution generator on the MI355X conv core.

Drop-in for ``ResidualBlock`` / ``GeneratorSR`` / ``GeneratorDown`` of /root/reference/CycleGAN/CycleGAN_w_v1/CycleGAN.py:89-163
(the v2 script, CycleGAN_w_v2/CycleGAN2.py:102-176, defines the same networks): same constructor arguments, same
``state_dict`` keys (``conv1.0.*``, ``res_blocks.N.conv{1,2}.*``, ``up{1,2}.0.*``, ``conv_out.*``), NCHW fp32 in and
out, trainable through autograd.  Every 3x3 convolution — the residual blocks and the two sub-pixel stages, 97 % of
the FLOPs — runs on the HIP kernels of ``conv_core`` (bf16 operands, fp32 accumulation) with ReLU fused as a
LeakyReLU of slope 0; the residual trunk is carried in fp32 beside the bf16 copy the next convolution reads, both written by
the epilogue of a block's second conv (which adds the skip); PixelShuffle is a HIP kernel on the blocked layout.  The
two 7x7 convolutions and tanh stay on stock PyTorch-ROCm.
"""
import torch
from torch import Tensor, nn

from . import conv_core as cc


def _need_cuda(x: Tensor, who: str) -> None:
    if not x.is_cuda:
        raise RuntimeError(f"super_resolution_amd.{who} runs on ROCm devices only: move the module and its input to "
                           "'cuda' (there is no CPU fallback)")
    if x.dim() != 4 or not x.is_floating_point():
        # (the reference's first nn.Conv2d refuses integer images the same way)
        raise RuntimeError(f"{who}: expected a floating-point NCHW tensor; Input type ({x.dtype}, shape {tuple(x.shape)}) "
                           "and weight type (torch.float32) should be the same")


def _nchw_to_f32(v: Tensor) -> Tensor:
    """NCHW fp32 -> the conv core's fp32 layout [N, H, C/4, W, 4]."""
    N, C, H, W = v.shape
    return v.view(N, C // 4, 4, H, W).permute(0, 3, 1, 4, 2).contiguous()


class ResidualBlock(nn.Module):
    """CycleGAN.py:89-100 — ``x + conv2(relu(conv1(x)))``."""

    def __init__(self, channels: int):
        super().__init__()
        if channels % 16:
            raise ValueError("channels must be a multiple of 16 (the fp32 trunk output runs on 16-channel groups)")
        self.conv1 = nn.Conv2d(channels, channels, kernel_size=3, padding=1)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(channels, channels, kernel_size=3, padding=1)

    def _run(self, xb: Tensor, trunk: Tensor):
        """(blocked bf16 x, fp32 x) -> the same pair for the block's output."""
        h = cc.conv3x3(xb, self.conv1.weight, self.conv1.bias, 0.0)              # ReLU fused
        # the skip is added in conv2's epilogue, which writes the fp32 trunk and its bf16 copy: one launch
        trunk, xb = cc.conv3x3_add(h, self.conv2.weight, self.conv2.bias, trunk)
        return xb, trunk

    def forward(self, x: Tensor) -> Tensor:
        _need_cuda(x, "ResidualBlock")
        v = x.float()
        _xb, trunk = self._run(cc.to_blocked(v), _nchw_to_f32(v))
        return cc.f32_as_nchw(trunk)


class GeneratorSR(nn.Module):
    """CycleGAN.py:103-137 — LR -> HR (x4): 7x7 conv + ReLU, residual blocks, two (3x3 conv to 4x channels,
    PixelShuffle(2), ReLU) stages, 7x7 conv, tanh."""

    def __init__(self, in_channels: int = 3, out_channels: int = 3, num_residual: int = 6, base_channels: int = 64):
        super().__init__()
        if base_channels % 16:
            raise ValueError("base_channels must be a multiple of 16 (the fp32 trunk output runs on 16-channel groups)")
        self.conv1 = nn.Sequential(nn.Conv2d(in_channels, base_channels, kernel_size=7, stride=1, padding=3),
                                   nn.ReLU(inplace=True))
        self.res_blocks = nn.Sequential(*[ResidualBlock(base_channels) for _ in range(num_residual)])
        self.up1 = nn.Sequential(nn.Conv2d(base_channels, base_channels * 4, kernel_size=3, stride=1, padding=1),
                                 nn.PixelShuffle(2), nn.ReLU(inplace=True))
        self.up2 = nn.Sequential(nn.Conv2d(base_channels, base_channels * 4, kernel_size=3, stride=1, padding=1),
                                 nn.PixelShuffle(2), nn.ReLU(inplace=True))
        self.conv_out = nn.Conv2d(base_channels, out_channels, kernel_size=7, stride=1, padding=3)
        self.tanh = nn.Tanh()

    def forward(self, x: Tensor) -> Tensor:
        _need_cuda(x, "GeneratorSR")
        v = self.conv1(x.float())                                   # 7x7: stock PyTorch-ROCm
        xb, trunk = cc.to_blocked(v), _nchw_to_f32(v)
        for block in self.res_blocks:
            xb, trunk = block._run(xb, trunk)
        for up in (self.up1, self.up2):
            conv = up[0]
            # ReLU commutes with the pixel shuffle: fused into the convolution as slope 0; the shuffle is one HIP
            # kernel on the blocked layout
            xb = cc.pixel_shuffle2(cc.conv3x3(xb, conv.weight, conv.bias, 0.0))
        return self.tanh(self.conv_out(cc.from_blocked(xb, self.conv_out.in_channels)))


class GeneratorDown(nn.Module):
    """CycleGAN.py:140-163 — HR -> LR (/4): 7x7 conv + ReLU, two (4x4 / stride 2 conv, ReLU) stages, two residual blocks
    at four times the base width, 7x7 conv, tanh.  The residual blocks' 3x3 convs run on the conv core (as in
    ``GeneratorSR``); the 7x7 and 4x4 convs stay on stock PyTorch-ROCm."""

    def __init__(self, in_channels: int = 3, out_channels: int = 3, base_channels: int = 64):
        super().__init__()
        if (4 * base_channels) % 16:
            raise ValueError("4 * base_channels must be a multiple of 16 (the fp32 trunk output runs on 16-channel groups)")
        b = base_channels
        self.down = nn.Sequential(nn.Conv2d(in_channels, b, kernel_size=7, stride=1, padding=3), nn.ReLU(inplace=True),
                                  nn.Conv2d(b, b * 2, kernel_size=4, stride=2, padding=1), nn.ReLU(inplace=True),
                                  nn.Conv2d(b * 2, b * 4, kernel_size=4, stride=2, padding=1), nn.ReLU(inplace=True))
        self.res_blocks = nn.Sequential(ResidualBlock(b * 4), ResidualBlock(b * 4))
        self.out_conv = nn.Conv2d(b * 4, out_channels, kernel_size=7, stride=1, padding=3)
        self.tanh = nn.Tanh()

    def forward(self, x: Tensor) -> Tensor:
        _need_cuda(x, "GeneratorDown")
        v = self.down(x.float())                                    # 7x7 and 4x4 / stride 2: stock PyTorch-ROCm
        xb, trunk = cc.to_blocked(v), _nchw_to_f32(v)
        for block in self.res_blocks:
            xb, trunk = block._run(xb, trunk)
        return self.tanh(self.out_conv(cc.f32_as_nchw(trunk)))
