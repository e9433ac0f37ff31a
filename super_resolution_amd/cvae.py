"""The conditional VAE on the MI355X conv core.

Drop-in for ``CVAE`` of /root/reference/condtional VAE/model.py:53-215: same constructor argument, the same
``state_dict`` keys (``enc_conv{1..4}``, ``fc_mu``, ``fc_logvar``, ``decoder_input``, ``condition_processor``,
``dec_conv{1..3}``, ``upsample{1,2}``, ``final_conv``, ``lr_upsampler``), the same ``encode`` / ``reparameterize`` /
``decode`` / ``forward`` / ``sample`` methods, trainable through autograd.  Its eight 3x3 convolutions that produce
something — the encoder (its stride-2 convs computed at the output positions, each with its BatchNorm + LeakyReLU as
``conv_core.conv_bn_act``), the two sub-pixel stages, the final conv and the conv behind the bicubic up-sampler — run
on the HIP kernels of ``conv_core`` (bf16 operands, fp32 accumulation; fp32 conv output into every BatchNorm).  The 4x4
transposed convolutions, the linear layers, the decoder's BatchNorms, PixelShuffle, bicubic interpolation and tanh stay
on stock PyTorch-ROCm.  ``condition_processor`` keeps its
parameters (checkpoints carry them) but is not evaluated: the reference computes its output and never uses it
(model.py:175).  ``VGGPerceptualLoss`` of the same file is below: its convs, pools and L1 reductions on the conv core too.
"""
import torch
from torch import Tensor, nn
import torch.nn.functional as F

from . import conv_core as cc


def _need_cuda(x: Tensor) -> None:
    if not x.is_cuda:
        raise RuntimeError("super_resolution_amd.cvae.CVAE runs on ROCm devices only: move the module and its input to "
                           "'cuda' (there is no CPU fallback)")
    if not x.is_floating_point():
        # (the reference's first nn.Conv2d refuses integer images the same way)
        raise RuntimeError(f"CVAE: Input type ({x.dtype}) and weight type (torch.float32) should be the same")


def _conv_f32(x_nchw: Tensor, conv: nn.Conv2d) -> Tensor:
    """3x3 / pad 1 conv (stride 1 or 2) of an NCHW fp32 tensor on the conv core, NCHW fp32 result."""
    xb = cc.to_blocked(x_nchw.contiguous())
    stride = int(conv.stride[0])
    if conv.out_channels % 16:
        return cc.conv3x3(xb, conv.weight, conv.bias, 1.0, nchw_out=True, stride=stride)
    return cc.f32_as_nchw(cc.conv3x3(xb, conv.weight, conv.bias, 1.0, f32_out=True, stride=stride))


class CVAE(nn.Module):
    def __init__(self, latent_dim: int = 256):
        super().__init__()
        self.latent_dim = latent_dim
        self.enc_conv1 = nn.Sequential(nn.Conv2d(3, 64, kernel_size=3, stride=1, padding=1), nn.LeakyReLU(0.2))
        self.enc_conv2 = nn.Sequential(nn.Conv2d(64, 128, kernel_size=3, stride=2, padding=1), nn.BatchNorm2d(128),
                                       nn.LeakyReLU(0.2))
        self.enc_conv3 = nn.Sequential(nn.Conv2d(128, 256, kernel_size=3, stride=2, padding=1), nn.BatchNorm2d(256),
                                       nn.LeakyReLU(0.2))
        self.enc_conv4 = nn.Sequential(nn.Conv2d(256, 512, kernel_size=3, stride=2, padding=1), nn.BatchNorm2d(512),
                                       nn.LeakyReLU(0.2))
        self.flatten = nn.Flatten()
        self.fc_mu = nn.Linear(512 * 4 * 4, latent_dim)
        self.fc_logvar = nn.Linear(512 * 4 * 4, latent_dim)
        self.decoder_input = nn.Linear(latent_dim, 512 * 4 * 4)
        self.condition_processor = nn.Sequential(nn.Conv2d(3, 32, kernel_size=3, stride=1, padding=1), nn.LeakyReLU(0.2),
                                                 nn.Conv2d(32, 64, kernel_size=3, stride=1, padding=1), nn.LeakyReLU(0.2))
        self.dec_conv1 = nn.Sequential(nn.ConvTranspose2d(512, 256, kernel_size=4, stride=2, padding=1),
                                       nn.BatchNorm2d(256), nn.LeakyReLU(0.2))
        self.dec_conv2 = nn.Sequential(nn.ConvTranspose2d(256 + 256, 128, kernel_size=4, stride=2, padding=1),
                                       nn.BatchNorm2d(128), nn.LeakyReLU(0.2))
        self.dec_conv3 = nn.Sequential(nn.ConvTranspose2d(128 + 128, 64, kernel_size=4, stride=2, padding=1),
                                       nn.BatchNorm2d(64), nn.LeakyReLU(0.2))
        self.upsample1 = nn.Sequential(nn.Conv2d(64 + 64, 256, kernel_size=3, stride=1, padding=1), nn.PixelShuffle(2),
                                       nn.BatchNorm2d(64), nn.LeakyReLU(0.2))
        self.upsample2 = nn.Sequential(nn.Conv2d(64, 256, kernel_size=3, stride=1, padding=1), nn.PixelShuffle(2),
                                       nn.BatchNorm2d(64), nn.LeakyReLU(0.2))
        self.final_conv = nn.Sequential(nn.Conv2d(64, 3, kernel_size=3, stride=1, padding=1), nn.Tanh())
        self.lr_upsampler = nn.Sequential(nn.Upsample(scale_factor=4, mode='bicubic', align_corners=False),
                                          nn.Conv2d(3, 3, kernel_size=3, stride=1, padding=1))

    def encode(self, x: Tensor):
        _need_cuda(x)
        x1 = F.leaky_relu(_conv_f32(x.float(), self.enc_conv1[0]), 0.2)
        feats, hb, h = [x1], cc.to_blocked(x1.contiguous()), x1
        for block in (self.enc_conv2, self.enc_conv3, self.enc_conv4):
            # stride-2 conv -> BatchNorm (batch statistics in training mode) -> LeakyReLU, blocked bf16 throughout
            hb = cc.conv_bn_act(hb, block[0], block[1], 0.2)
            h = cc.from_blocked(hb, block[0].out_channels)
            feats.append(h)
        encoded = self.flatten(h)
        return self.fc_mu(encoded), self.fc_logvar(encoded), tuple(feats)

    def reparameterize(self, mu: Tensor, logvar: Tensor) -> Tensor:
        std = torch.exp(0.5 * logvar)
        return mu + torch.randn_like(std) * std

    def decode(self, z: Tensor, condition: Tensor, skip_features) -> Tensor:
        _need_cuda(condition)
        skip1, skip2, skip3, _skip4 = skip_features
        x = self.decoder_input(z).view(-1, 512, 4, 4)
        x = torch.cat([self.dec_conv1(x), skip3], dim=1)           # 4x4 transposed convs: stock PyTorch-ROCm
        x = torch.cat([self.dec_conv2(x), skip2], dim=1)
        x = torch.cat([self.dec_conv3(x), skip1], dim=1)
        for block in (self.upsample1, self.upsample2):
            x = F.leaky_relu(block[2](F.pixel_shuffle(_conv_f32(x, block[0]), 2)), 0.2)
        x = torch.tanh(_conv_f32(x, self.final_conv[0]))
        upsampled_lr = _conv_f32(self.lr_upsampler[0](condition.float()), self.lr_upsampler[1])
        return x + upsampled_lr

    def forward(self, x: Tensor):
        mu, logvar, skip_features = self.encode(x)
        z = self.reparameterize(mu, logvar)
        return self.decode(z, x, skip_features), mu, logvar

    def sample(self, num_samples: int, condition: Tensor) -> Tensor:
        z = torch.randn(num_samples, self.latent_dim, device=condition.device)
        b, dev = condition.size(0), condition.device
        dummy = (torch.zeros(b, 64, 32, 32, device=dev), torch.zeros(b, 128, 16, 16, device=dev),
                 torch.zeros(b, 256, 8, 8, device=dev), torch.zeros(b, 512, 4, 4, device=dev))
        return self.decode(z, condition, dummy)


class VGGPerceptualLoss(nn.Module):
    """``VGGPerceptualLoss`` of /root/reference/condtional VAE/model.py:8-50: the sum over four VGG19 stages (relu1_2,
    relu2_2, relu3_4, relu4_4 = ``features[:4]``, ``[4:9]``, ``[9:18]``, ``[18:27]``) of the mean L1 distance between
    the features of two images given in [-1, 1].  Same ``state_dict`` keys (``blocks.K.N.weight`` with torchvision's
    layer numbers, buffers ``mean`` / ``std``), parameters frozen.  The twelve 3x3 convs (ReLU fused), the three max
    pools and the four L1 reductions run on the HIP kernels of ``conv_core``; the gradient flows to the first argument
    (and to the second where it requires one), as in the reference.

    The ImageNet weights come from torchvision as in the reference (:11); where torchvision is not installed pass
    ``pretrained=False`` and load a ``state_dict`` with the reference's keys."""

    def __init__(self, *, pretrained: bool = True):
        super().__init__()
        from .esrgan_aux import _vgg19_feature_layers
        layers = _vgg19_feature_layers()[:27]
        if pretrained:
            try:
                from torchvision import models
            except ImportError as exc:
                raise ImportError("VGGPerceptualLoss(pretrained=True) takes the ImageNet weights from torchvision, which "
                                  "is not installed; use pretrained=False and load_state_dict") from exc
            for dst, src in zip(layers, models.vgg19(pretrained=True).features):
                if isinstance(dst, nn.Conv2d):
                    dst.load_state_dict(src.state_dict())
        features = nn.Sequential(*layers)
        self.blocks = nn.ModuleList([features[:4], features[4:9], features[9:18], features[18:27]])   # (keys keep the
        for p in self.parameters():                                                                # original numbers)
            p.requires_grad = False
        self.register_buffer("mean", torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1))
        self.register_buffer("std", torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1))

    def normalize(self, x: Tensor) -> Tensor:
        return (x - self.mean) / self.std

    @staticmethod
    def _run_block(block: nn.Sequential, h: Tensor) -> Tensor:
        mods = list(block)
        i = 0
        while i < len(mods):
            m = mods[i]
            i += 1
            if isinstance(m, nn.Conv2d):
                relu = i < len(mods) and isinstance(mods[i], nn.ReLU)
                h = cc.conv3x3(h, m.weight, m.bias, 0.0 if relu else 1.0)     # ReLU = LeakyReLU of slope 0, fused
                i += int(relu)
            elif isinstance(m, nn.MaxPool2d):
                h = cc.max_pool2x2(h)
            else:
                h = torch.relu(h)
        return h

    def forward(self, x: Tensor, y: Tensor) -> Tensor:
        _need_cuda(x)
        xb = cc.to_blocked(self.normalize((x.float() + 1) / 2).contiguous())
        yb = cc.to_blocked(self.normalize((y.float() + 1) / 2).contiguous())
        loss = 0.0
        for block in self.blocks:
            xb, yb = self._run_block(block, xb), self._run_block(block, yb)
            # (VGG19's widths are whole units: the mean over the blocked tensor is the mean over the NCHW one)
            loss = loss + cc.l1_loss(xb, yb)
        return loss
