"""GPU: the conv core on randomly drawn shapes (seeded): forward, data, weight and bias gradients of one 3x3 conv through
``conv_core.conv3x3`` (what the discriminator, VGG19, CycleGAN and CVAE layers call; the kernels of
/root/reference/esrgan/models/esrgan_model.py:10-14 and :67-84) against torch's conv2d in fp64 on the same bf16-rounded
operands.  The fixed cases of test_gpu_parity.py / test_gpu_backward.py pin the shapes the networks use; this sweep walks
the kernel-selection boundaries in between (image sizes around the tile sizes, channel counts around the unit and block
sizes, both strides, fused activations)."""
import random

import pytest
import torch
import torch.nn.functional as F

from super_resolution_amd import conv_core as cc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def bf16r(t):
    return t.to(torch.bfloat16).float()


def rel_l2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def draw_cases(n, seed):
    rng = random.Random(seed)
    cases = []
    for _ in range(n):
        cin = rng.choice([3, 8, 16, 24, 32, 64, 96, 160, 192])
        cout = rng.choice([4, 8, 16, 32, 48, 64, 96, 128])
        H = rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 33, 40])
        W = rng.choice([1, 2, 3, 5, 8, 16, 31, 32, 33, 34, 45, 63, 64, 65])
        cases.append((rng.randint(1, 3), cin, cout, H, W, rng.choice([1, 1, 2]), rng.choice([1.0, 0.2, 0.0])))
    return cases


# output widths that are neither a 32-channel block nor within one 64-channel launch of the weight gradient (its last slice
# is ragged), and input widths that are not whole 32-channel chunks
RAGGED = [(2, 40, 72, 9, 33, 1, 0.2), (1, 64, 80, 17, 16, 1, 1.0), (2, 24, 200, 5, 8, 2, 0.2), (1, 104, 136, 8, 31, 1, 0.0),
          (3, 8, 88, 4, 65, 2, 1.0)]


@pytest.mark.parametrize("case", draw_cases(36, 20240607) + RAGGED, ids=lambda c: "n%d_ci%d_co%d_%dx%d_s%d_a%g" % c)
def test_conv_forward_and_gradients_on_random_shapes(case):
    N, cin, cout, H, W, stride, slope = case
    g = torch.Generator().manual_seed(hash(case) & 0xffff)
    x = torch.randn(N, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    xd = x.to(DEV).requires_grad_(True)
    wd = w.to(DEV).requires_grad_(True)
    bd = b.to(DEV).requires_grad_(True)
    y = cc.from_blocked(cc.conv3x3(cc.to_blocked(xd), wd, bd, slope, stride=stride), cout)
    Ho, Wo = ((H + 1) // 2, (W + 1) // 2) if stride == 2 else (H, W)
    assert y.shape == (N, cout, Ho, Wo)
    r = torch.randn(N, cout, Ho, Wo, generator=g)
    (y * r.to(DEV)).sum().backward()

    xr = bf16r(x).double().requires_grad_(True)
    wr = bf16r(w).double().requires_grad_(True)
    br = b.double().requires_grad_(True)
    pre = F.conv2d(xr, wr, br, padding=1, stride=stride)
    ref = torch.where(pre > 0, pre, pre * slope)
    # the backward reads the cotangent as bf16 and gates it by the sign of the bf16-rounded output
    (ref * bf16r(r).double()).sum().backward()
    scale = max(1.0, ref.abs().max().item())
    assert (y.detach().cpu().double() - ref.detach()).abs().max().item() < 1e-2 * scale      # one bf16 rounding of the output
    assert rel_l2(wd.grad.cpu(), wr.grad) < 2e-2 and rel_l2(bd.grad.cpu(), br.grad) < 2e-2
    # x.grad passes through bf16 storage once more (the blocked data gradient)
    assert rel_l2(xd.grad.cpu(), xr.grad) < 2e-2
