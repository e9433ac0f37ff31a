"""GPU: the CycleGAN super-resolution generator on the conv core against the CPU oracle and the vector made from
the reference class (CycleGAN/CycleGAN_w_v1/CycleGAN.py:89-137).  3x3 convolutions take bf16 operands (fp32
accumulation, fp32 residual trunk); outputs are tanh values in [-1, 1], compared by PSNR at peak 2 and by
maximum error; gradients by relative L2 per tensor."""
import numpy as np
import pytest
import torch

from oracle import esrgan_oracle as O
from super_resolution_amd.cyclegan import GeneratorSR, ResidualBlock

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel_l2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def psnr2(a, b):
    return (10 * torch.log10(4.0 / (a.double() - b.double()).pow(2).mean())).item()


def test_generator_sr_against_reference_vector(golden):
    g = golden("g9_cyclegan_sr.npz")
    sd = O.make_generator_sr_state(int(g["seed"]), num_residual=int(g["num_residual"]))
    net = GeneratorSR(num_residual=int(g["num_residual"]))
    net.load_state_dict(sd)
    net = net.to(DEV)
    x = torch.from_numpy(g["x"]).to(DEV).requires_grad_(True)
    out = net(x)
    ref = torch.from_numpy(g["out"])
    assert out.shape == ref.shape and out.dtype == torch.float32
    p = psnr2(out.detach().cpu(), ref)
    err = (out.detach().cpu() - ref).abs().max().item()
    print(f"GeneratorSR bf16 vs reference fp32: PSNR {p:.1f} dB (peak 2), max |err| {err:.4f}")
    assert p >= 50.0 and err < 2e-2
    (out * torch.from_numpy(g["r"]).to(DEV)).sum().backward()
    # a white-noise cotangent through 8 ReLU layers: bf16 rounding of the forward flips ReLU gates of
    # near-zero pre-activations, which moves individual gradient entries by more than the rounding itself
    # (the same effect bounds the ESRGAN generator's white-noise test); sums and directions stay close
    gx = rel_l2(x.grad.cpu(), torch.from_numpy(g["grad_x"]))
    worst = 0.0
    for name, prm in net.named_parameters():
        got = prm.grad.double().abs().sum().item()
        want = float(g[f"gabs/{name}"])
        worst = max(worst, abs(got - want) / want)
    full = {name: rel_l2(dict(net.named_parameters())[name].grad.cpu(), torch.from_numpy(g[f"gfull/{name}"]))
            for name in ("conv1.0.bias", "res_blocks.2.conv2.bias", "conv_out.bias")}
    print(f"grad_x rel L2 {gx:.3f}; worst |grad| sum deviation {worst:.3f}; bias grads {full}")
    assert gx < 0.12 and worst < 0.05
    assert all(v < 0.12 for v in full.values()), full


def test_generator_sr_default_config_and_block():
    """The default network (6 residual blocks) on a ragged batch, and one residual block on its own."""
    sd = O.make_generator_sr_state(11)
    net = GeneratorSR()
    net.load_state_dict(sd)
    net = net.to(DEV).eval()
    x = torch.rand(3, 3, 21, 45, generator=torch.Generator().manual_seed(12)) * 2 - 1
    with torch.no_grad():
        out = net(x.to(DEV)).cpu()
        ref = O.generator_sr_forward(sd, x)
    assert out.shape == (3, 3, 84, 180)
    assert psnr2(out, ref) >= 50.0
    blk = ResidualBlock(64).to(DEV)
    xb = torch.randn(2, 64, 9, 13, generator=torch.Generator().manual_seed(13))
    with torch.no_grad():
        got = blk(xb.to(DEV)).cpu()
        F = torch.nn.functional
        w1, b1, w2, b2 = (t.detach().cpu() for t in (blk.conv1.weight, blk.conv1.bias, blk.conv2.weight, blk.conv2.bias))
        want = xb + F.conv2d(F.relu(F.conv2d(xb, w1, b1, padding=1)), w2, b2, padding=1)
    assert rel_l2(got, want) < 5e-3
