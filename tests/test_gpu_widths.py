"""GPU: constructor arguments other than the defaults.  The reference's generator takes any channel counts
(/root/reference/esrgan/models/esrgan_model.py:42-53: in_channels, out_channels, num_features, num_blocks,
growth_channels); the fused kernels here are built for 64 features and a growth of 32, everything else runs on the
general conv kernels.  Forward in both precisions against the oracle, and one recorded forward + backward of the bf16
path like for like with the oracle's gradients."""
import pytest
import torch

from oracle import esrgan_oracle as O
import super_resolution_amd as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# (in_channels, out_channels, num_features, num_blocks, growth_channels)
CONFIGS = [(1, 1, 32, 2, 16), (4, 2, 48, 1, 16), (3, 3, 64, 1, 16), (3, 3, 32, 1, 32), (3, 3, 128, 1, 32),
           (3, 3, 64, 1, 64), (2, 5, 96, 1, 48), (3, 3, 16, 1, 16), (3, 3, 64, 0, 32), (3, 3, 80, 1, 32), (3, 7, 112, 1, 80)]


def rel_l2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def state_of(cfg, seed):
    i, o, nf, nb, gc = cfg
    return O.make_generator_state(seed, in_channels=i, out_channels=o, num_features=nf, num_blocks=nb,
                                  growth_channels=gc)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("cfg", CONFIGS)
def test_forward_with_other_channel_counts_matches_oracle(cfg, precision):
    sd = state_of(cfg, 301)
    x = torch.rand(2, cfg[0], 20, 36, generator=torch.Generator().manual_seed(302))
    net = S.GeneratorRRDB(*cfg, precision=precision)
    assert list(net.state_dict().keys()) == list(sd.keys())
    net.load_state_dict(sd)
    net = net.to(DEV).eval()
    with torch.no_grad():
        out = net(x.to(DEV)).cpu()
        ref = O.generator_forward(sd, x)
    assert out.shape == (2, cfg[1], 80, 144)
    if precision == "fp32":
        assert (out - ref).abs().max().item() <= 1e-4
    else:
        assert O.psnr(out, ref) >= 50.0


@pytest.mark.parametrize("cfg", CONFIGS)
def test_training_pass_with_other_channel_counts_like_for_like(cfg):
    """Every parameter gradient and the input gradient under the pixel loss (train.py:160) against the oracle run with
    the HIP path's rounding points: <= 6 % relative L2 per tensor."""
    sd = state_of(cfg, 311)
    gen = torch.Generator().manual_seed(312)
    x0 = torch.rand(2, cfg[0], 12, 20, generator=gen)
    target = torch.rand(2, cfg[1], 48, 80, generator=gen)
    net = S.GeneratorRRDB(*cfg)
    net.load_state_dict(sd)
    net = net.to(DEV).train()
    x = x0.to(DEV).requires_grad_(True)
    torch.nn.functional.l1_loss(net(x), target.to(DEV)).backward()
    with torch.no_grad():
        out_ref = O.generator_forward(sd, x0, emulate_bf16="hr_split")
    r = torch.sign(out_ref - target) / out_ref.numel()
    _o, ref_grads, ref_gx = O.generator_grads(sd, x0, r, input_grad=True, emulate_bf16="hr_split")
    errs = sorted(((rel_l2(p.grad.cpu(), ref_grads[k]), k) for k, p in net.named_parameters()), reverse=True)
    assert len(errs) == len(sd)
    assert errs[0][0] < 6e-2, errs[:3]
    assert rel_l2(x.grad.cpu(), ref_gx) < 6e-2


@pytest.mark.parametrize("cfg", [(1, 1, 32, 1, 16), (3, 3, 128, 1, 32), (2, 5, 96, 1, 48)])
def test_fp32_training_pass_with_other_channel_counts(cfg):
    """The exact-fp32 path on the same widths: every gradient within 2e-5 relative L2 of the oracle's.  (Seeded so that no
    pre-activation lies within fp32 rounding of zero: weights from seed 321 at 128 features have one, the two runs then
    take different LeakyReLU slopes for that element, and every gradient below it differs by 3e-4.)"""
    sd = state_of(cfg, 323)
    gen = torch.Generator().manual_seed(322)
    x0 = torch.rand(1, cfg[0], 9, 11, generator=gen)
    r = torch.randn(1, cfg[1], 36, 44, generator=gen)
    net = S.GeneratorRRDB(*cfg, precision="fp32")
    net.load_state_dict(sd)
    net = net.to(DEV).train()
    x = x0.to(DEV).requires_grad_(True)
    net(x).backward(r.to(DEV))
    _o, ref_grads, ref_gx = O.generator_grads(sd, x0, r, input_grad=True)
    errs = sorted(((rel_l2(p.grad.cpu(), ref_grads[k]), k) for k, p in net.named_parameters()), reverse=True)
    assert errs[0][0] < 2e-5, errs[:3]
    assert rel_l2(x.grad.cpu(), ref_gx) < 2e-5


@pytest.mark.parametrize("widths", [(32, 16), (48, 16), (64, 64), (128, 32), (96, 48), (16, 16), (80, 32), (112, 80)])
@pytest.mark.parametrize("whole_rrdb", [False, True])
def test_stand_alone_blocks_with_other_channel_counts(widths, whole_rrdb):
    """ResidualDenseBlock(in_channels, growth_channels) and RRDB on their own (esrgan_model.py:8-39): forward in both
    precisions, and the bf16 backward against autograd through the oracle run with the same rounding points."""
    nf, gc = widths
    names = [f"rdb{r}." for r in (1, 2, 3)] if whole_rrdb else [""]
    specs = [(f"{p}conv{k + 1}", nf + gc * k, gc if k < 4 else nf) for p in names for k in range(5)]
    sd = O.make_conv_state(specs, torch.Generator().manual_seed(331))
    gen = torch.Generator().manual_seed(332)
    x0 = torch.randn(2, nf, 10, 14, generator=gen)
    r = torch.randn(2, nf, 10, 14, generator=gen)
    fwd = (lambda s, x, e=False: O.rrdb_forward(s, "", x, e)) if whole_rrdb else \
          (lambda s, x, e=False: O.dense_block_forward(s, "", x, e))
    with torch.no_grad():
        ref = fwd(sd, x0)
    for precision, tol in (("fp32", 3e-5), ("bf16", 4e-2)):
        net = (S.RRDB if whole_rrdb else S.ResidualDenseBlock)(nf, gc, precision=precision)
        net.load_state_dict(sd)
        net = net.to(DEV)
        with torch.no_grad():
            out = net(x0.to(DEV)).cpu()
        assert (out - ref).abs().max().item() < tol * max(1.0, ref.abs().max().item()), precision
    net.train()
    x = x0.to(DEV).requires_grad_(True)
    net(x).backward(r.to(DEV))
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xin = x0.clone().requires_grad_(True)
    fwd(leaves, xin, True).backward(r)
    errs = sorted(((rel_l2(p.grad.cpu(), leaves[k].grad), k) for k, p in net.named_parameters()), reverse=True)
    assert len(errs) == len(sd) and errs[0][0] < 6e-2, errs[:3]
    assert rel_l2(x.grad.cpu(), xin.grad) < 6e-2
