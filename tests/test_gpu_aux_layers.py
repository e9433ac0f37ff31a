"""GPU: the layers around the discriminator's / VGG19's convs on the blocked layout — stride-2 conv, BatchNorm +
LeakyReLU, 2x2 max pooling — and the mean L1 / MSE reductions, each against the torch op it replaces
(/root/reference/esrgan/models/esrgan_model.py:67-84, :93-98; /root/reference/esrgan/train.py:114-115), forward and
backward.  bf16 operands are rounded on both sides, so the tolerances cover summation order and the final bf16
store only."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from super_resolution_amd import conv_core as cc
from super_resolution_amd import ops
from super_resolution_amd._lib import DTYPE_BF16

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def bf16r(x):
    return x.to(torch.bfloat16).to(torch.float32)


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


STRIDE2_CASES = [(2, 64, 64, 16, 32), (1, 64, 128, 9, 37), (1, 128, 128, 33, 31), (3, 32, 32, 8, 8), (1, 64, 64, 1, 1)]


@pytest.mark.parametrize("case", STRIDE2_CASES)
def test_stride2_conv_forward_and_gradients(case):
    N, cin, cout, H, W = case
    x = rnd(N, cin, H, W, seed=1)
    w = rnd(cout, cin, 3, 3, seed=2, scale=0.05)
    b = rnd(cout, seed=3, scale=0.1)
    r = rnd(N, cout, (H + 1) // 2, (W + 1) // 2, seed=4)
    # reference: fp64 conv on bf16-rounded operands
    xr = bf16r(x).double().requires_grad_(True)
    wr = bf16r(w).double().requires_grad_(True)
    br = b.double().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, stride=2, padding=1)
    (yr * bf16r(r).double()).sum().backward()
    # HIP
    xg = x.to(DEV).requires_grad_(True)
    wg = w.to(DEV).requires_grad_(True)
    bg = b.to(DEV).requires_grad_(True)
    yb = cc.conv3x3(cc.to_blocked(xg), wg, bg, 1.0, stride=2)
    assert yb.shape[1] == (H + 1) // 2 and yb.shape[3] == (W + 1) // 2
    y = cc.from_blocked(yb, cout)
    (y * r.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    assert rel(y.detach().cpu(), yr.detach()) < 6e-3            # bf16 store of the output
    assert rel(xg.grad.cpu(), xr.grad) < 6e-3                    # bf16 gradient stores
    assert rel(wg.grad.cpu(), wr.grad) < 6e-3
    assert rel(bg.grad.cpu(), br.grad) < 6e-3
    # fp32 output kind: no rounding of the result
    if cout % 16 == 0:
        y32 = cc.f32_as_nchw(cc.conv3x3(cc.to_blocked(x.to(DEV)), w.to(DEV), b.to(DEV), 1.0, f32_out=True, stride=2))
        assert (y32.cpu().double() - yr.detach()).abs().max().item() < 2e-4 * max(1.0, yr.abs().max().item())


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("case", [(4, 64, 64, 16, 16, 2), (2, 64, 128, 12, 20, 1), (2, 128, 128, 7, 9, 2)])
def test_conv_batchnorm_leakyrelu_matches_torch(case, training):
    N, cin, cout, H, W, stride = case
    torch.manual_seed(5)
    conv = nn.Conv2d(cin, cout, 3, stride, 1)
    bn = nn.BatchNorm2d(cout)
    with torch.no_grad():   # a normalisation that is not at its initial state
        bn.weight.copy_(1.0 + 0.3 * torch.randn(cout))
        bn.bias.copy_(0.2 * torch.randn(cout))
        bn.running_mean.copy_(0.1 * torch.randn(cout))
        bn.running_var.copy_(1.0 + 0.2 * torch.rand(cout))
        conv.weight.copy_(bf16r(conv.weight))
    conv_g, bn_g = nn.Conv2d(cin, cout, 3, stride, 1), nn.BatchNorm2d(cout)
    conv_g.load_state_dict(conv.state_dict())
    bn_g.load_state_dict(bn.state_dict())
    conv_g, bn_g = conv_g.to(DEV), bn_g.to(DEV)
    for m in (conv, bn, conv_g, bn_g):
        m.train(training)
    x = bf16r(rnd(N, cin, H, W, seed=6))
    r = rnd(N, cout, (H + stride - 1) // stride, (W + stride - 1) // stride, seed=7)
    xr = x.clone().requires_grad_(True)
    yr = F.leaky_relu(bn(conv(xr)), 0.2)
    (yr * r).sum().backward()
    xg = x.to(DEV).requires_grad_(True)
    yb = cc.conv_bn_act(cc.to_blocked(xg), conv_g, bn_g, 0.2)
    y = cc.from_blocked(yb, cout)
    (y * r.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    assert rel(y.detach().cpu(), yr.detach()) < 6e-3
    # the running buffers move (or stay) exactly as nn.BatchNorm2d's
    assert torch.allclose(bn_g.running_mean.cpu(), bn.running_mean, rtol=1e-4, atol=1e-5)
    assert torch.allclose(bn_g.running_var.cpu(), bn.running_var, rtol=1e-4, atol=1e-5)
    assert int(bn_g.num_batches_tracked) == int(bn.num_batches_tracked)
    # gradients: the incoming gradient and dx are bf16 stores; LeakyReLU signs of pre-activations within bf16 rounding
    # of zero may differ, hence the looser bound on the data path
    assert rel(bn_g.bias.grad.cpu(), bn.bias.grad) < 1e-2
    assert rel(bn_g.weight.grad.cpu(), bn.weight.grad) < 1e-2
    assert rel(conv_g.weight.grad.cpu(), conv.weight.grad) < 2e-2
    assert rel(xg.grad.cpu(), xr.grad) < 2e-2


@pytest.mark.parametrize("shape", [(2, 16, 8, 8), (1, 64, 9, 7), (3, 8, 2, 2), (1, 24, 5, 33)])
def test_max_pool_forward_and_first_maximum_gradient(shape):
    N, C, H, W = shape
    # few distinct values: ties in most windows
    x = torch.randint(0, 3, (N, C, H, W), generator=torch.Generator().manual_seed(8)).float()
    r = rnd(N, C, H // 2, W // 2, seed=9)
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 2, 2)
    (yr * bf16r(r)).sum().backward()
    xg = x.to(DEV).requires_grad_(True)
    yb = cc.max_pool2x2(cc.to_blocked(xg))
    y = cc.from_blocked(yb, C)
    (y * r.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    assert torch.equal(y.detach().cpu(), yr.detach())
    assert torch.equal(xg.grad.cpu(), bf16r(xr.grad))


@pytest.mark.parametrize("n", [1, 5, 4096, 16 * 3 * 128 * 128, 16 * 512 * 8 * 8 + 3])
def test_mean_losses_and_their_gradients(n):
    a, b = rnd(n, seed=10), rnd(n, seed=11)
    for fn, ref in ((cc.l1_loss, F.l1_loss), (cc.mse_loss, F.mse_loss)):
        ar = a.clone().double().requires_grad_(True)
        lr_ = ref(ar, b.double())
        (3.0 * lr_).backward()
        ag = a.to(DEV).requires_grad_(True)
        lg = fn(ag, b.to(DEV))
        (3.0 * lg).backward()
        torch.cuda.synchronize()
        assert abs(lg.item() - lr_.item()) < 1e-5 * max(1.0, abs(lr_.item()))
        assert (ag.grad.cpu().double() - ar.grad).abs().max().item() < 1e-6 * max(1.0, ar.grad.abs().max().item()) + 1e-12
    # constant targets (the adversarial terms' ones / zeros) and determinism
    ag = a.to(DEV).requires_grad_(True)
    l1 = cc.mse_loss(ag, 1.0)
    l2 = cc.mse_loss(ag, 1.0)
    assert l1.item() == l2.item()
    assert abs(l1.item() - F.mse_loss(a.double(), torch.ones(n, dtype=torch.float64)).item()) < 1e-5 * max(1.0, l1.item())


@pytest.mark.parametrize("case", [(4, 2, 64, 16, 16), (6, 3, 16, 5, 9), (2, 2, 512, 2, 2), (32, 2, 64, 64, 64), (8, 1, 32, 7, 3)])
@pytest.mark.parametrize("dy_f32", [False, True])
def test_grouped_batchnorm_is_the_separate_calls(case, dy_f32):
    """``groups`` consecutive sub-batches in one set of launches: outputs, statistics, running buffers (updated group by
    group) and every gradient bit-identical to one call per sub-batch (the discriminator update's two passes,
    /root/reference/esrgan/train.py:200-201, run as one batch)."""
    from super_resolution_amd import ops
    N, groups, C, H, W = case
    g = torch.Generator().manual_seed(N * 1000 + C)
    x = (torch.randn(N, H, C // 4, W, 4, generator=g) * 2 + 0.5).to(DEV)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(DEV), torch.randn(C, generator=g).to(DEV)
    part = N // groups

    def run(grouped):
        rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
        nb = torch.zeros((), dtype=torch.long, device=DEV)
        mean, rstd = torch.empty(groups, C, device=DEV), torch.empty(groups, C, device=DEV)
        out = torch.empty((N, H, C // 8, W, 8), dtype=torch.bfloat16, device=DEV)
        if dy_f32:
            dy = torch.randn(N, H, C // 4, W, 4, generator=torch.Generator().manual_seed(7)).to(DEV)
        else:
            dy = torch.randn(N, H, C // 8, W, 8, generator=torch.Generator().manual_seed(7)).to(DEV).bfloat16()
        dx = torch.empty_like(out)
        dgamma, dbeta = torch.empty(groups, C, device=DEV), torch.empty(groups, C, device=DEV)
        if grouped:
            ops.bn_lrelu_fwd(x, gamma, beta, 1e-5, 0.2, True, 0.1, rm, rv, nb, mean, rstd, out, 0, groups)
            ops.bn_lrelu_bwd(x, dy, gamma, beta, mean, rstd, 0.2, True, dx, dgamma, dbeta, 0, 0, groups)
        else:
            for k in range(groups):
                sl = slice(k * part, (k + 1) * part)
                ops.bn_lrelu_fwd(x[sl], gamma, beta, 1e-5, 0.2, True, 0.1, rm, rv, nb, mean[k], rstd[k], out[sl], 0)
                ops.bn_lrelu_bwd(x[sl], dy[sl], gamma, beta, mean[k], rstd[k], 0.2, True, dx[sl], dgamma[k], dbeta[k])
        return out, mean, rstd, rm, rv, nb, dx, dgamma, dbeta

    for a, b in zip(run(True), run(False)):
        assert torch.equal(a, b)
    with pytest.raises(RuntimeError, match="groups"):
        mean = torch.empty(groups, C, device=DEV)
        ops.bn_lrelu_fwd(x, gamma, beta, 1e-5, 0.2, True, 0.1, None, None, None, mean, mean.clone(),
                         torch.empty((N, H, C // 8, W, 8), dtype=torch.bfloat16, device=DEV), 0, N + 1)


def test_batched_weight_packs_are_the_single_packs():
    """sr355_pack_weights / sr355_pack_weights_transposed (all convs of a network in one launch each) write the bytes of
    the per-conv calls, and conv_core.prepare_packs installs them where the per-conv path looks."""
    from super_resolution_amd import conv_core as cc, ops
    from super_resolution_amd._lib import DTYPE_BF16
    g = torch.Generator().manual_seed(5)
    shapes = [(64, 3), (64, 64), (128, 64), (1, 512), (48, 24), (136, 104)]
    ws = [torch.randn(co, ci, 3, 3, generator=g).to(DEV) for co, ci in shapes]
    bs = [torch.randn(co, generator=g).to(DEV) for co, _ci in shapes]
    for (wp, bp), w, b in zip(ops.pack_weights_batch(ws, bs, DTYPE_BF16), ws, bs):
        wp1, bp1 = ops.pack_weight(w, b, DTYPE_BF16)
        assert torch.equal(wp[:wp1.numel()], wp1) and torch.equal(bp, bp1)
    for (wt, zb), w in zip(ops.pack_weights_transposed_batch(ws, DTYPE_BF16), ws):
        cout, cin = w.shape[0], w.shape[1]
        units, n_out = cc.units_for(cout), (cin + 3) // 4 * 4
        one = torch.zeros(units * 9 * ((n_out + 31) // 32 * 32) * 16, dtype=torch.uint8, device=DEV)
        ops.pack_weight_transposed(w, 0, cin, 1.0, one, units, 0, DTYPE_BF16)
        assert torch.equal(wt[:one.numel()], one) and not bool(zb.any()) and zb.numel() == (n_out + 31) // 32 * 32
    convs = [torch.nn.Conv2d(ci, co, 3, 1, 1).to(DEV) for co, ci in shapes[:4]]
    cc.prepare_packs(convs, True)
    for c in convs:
        packs = cc._packs_of(c.weight)
        assert packs.fwd is not None and packs.bwd is not None
        wp1, bp1 = ops.pack_weight(c.weight.detach(), c.bias.detach(), DTYPE_BF16)
        got_w, got_b = packs.forward_pack(c.weight, c.bias)
        assert torch.equal(got_w[:wp1.numel()], wp1) and torch.equal(got_b, bp1)
    with torch.no_grad():
        convs[1].weight.mul_(0.5)                      # a changed parameter is repacked, the others keep their packs
    kept = cc._packs_of(convs[0].weight).fwd
    cc.prepare_packs(convs, False)
    assert cc._packs_of(convs[0].weight).fwd is kept
    wp1, _ = ops.pack_weight(convs[1].weight.detach(), convs[1].bias.detach(), DTYPE_BF16)
    assert torch.equal(cc._packs_of(convs[1].weight).forward_pack(convs[1].weight, convs[1].bias)[0][:wp1.numel()], wp1)
