"""GPU: the tight gradient net.  The other training tests compare the bf16 HIP path with the fp32 reference run,
where bf16 rounding of near-zero pre-activations (LeakyReLU slope flips, BatchNorm residuals) forces loose bounds.
Here every comparison is like for like: the CPU oracle's forward rounds at the points where the HIP path rounds
(``emulate_bf16``; the casts pass gradients through), its backward is exact autograd, and every parameter tensor is
held to a few per cent — a wrong slice of one weight gradient cannot hide.  Also: stand-alone ResidualDenseBlock /
RRDB training against the reference's G1 / G2 gradients, GradScaler + FlatAdam, and the adversarial iteration at the
real 16 x 128 x 128 shape as a property test.  References: /root/reference/esrgan/models/esrgan_model.py:18-62,
:64-87; /root/reference/esrgan/train.py:158-163, :196-223."""
import numpy as np
import pytest
import torch

from oracle import esrgan_oracle as O
import super_resolution_amd as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def t(a):
    return torch.from_numpy(np.asarray(a))


def rel_l2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def worst_error(named_grads, ref):
    errs = sorted(((rel_l2(g.cpu(), ref[k]), k) for k, g in named_grads), reverse=True)
    return errs[0], errs


def test_full_depth_backward_every_tensor_like_for_like():
    """All 23 RRDBs, all 702 parameter gradients and the input gradient against the oracle run with the HIP path's
    rounding points (bf16 conv operands, split HR side): <= 5 % relative L2 per tensor."""
    sd = O.center_outputs(O.make_generator_state(71, num_blocks=23), 0.5)
    gen = torch.Generator().manual_seed(72)
    x0 = torch.rand(1, 3, 16, 16, generator=gen)
    target = torch.rand(1, 3, 64, 64, generator=gen)
    net = S.GeneratorRRDB(num_blocks=23)
    net.load_state_dict(sd)
    net = net.to(DEV).train()
    x = x0.to(DEV).requires_grad_(True)
    out = net(x)
    # the pixel loss of the pre-training phase (train.py:160): its cotangent is sign(out - target) / n
    loss = torch.nn.functional.l1_loss(out, target.to(DEV))
    loss.backward()
    with torch.no_grad():
        out_ref = O.generator_forward(sd, x0, emulate_bf16="hr_split")
    r = torch.sign(out_ref - target) / out_ref.numel()
    _o, ref_grads, ref_gx = O.generator_grads(sd, x0, r, input_grad=True, emulate_bf16="hr_split")
    (worst, name), errs = worst_error(((k, p.grad) for k, p in net.named_parameters()), ref_grads)
    print("full depth, like for like: worst", worst, name, " median", errs[len(errs) // 2][0])
    assert len(errs) == 702
    assert worst < 5e-2, (worst, name)
    assert rel_l2(x.grad.cpu(), ref_gx) < 5e-2


@pytest.mark.parametrize("shape", [(1, 5, 7), (2, 9, 32), (3, 32, 32), (1, 33, 31), (2, 16, 40), (1, 40, 33), (4, 12, 32),
                                   (1, 1, 1), (2, 2, 33)])
def test_training_pass_on_patch_shapes_around_the_kernel_boundaries(shape):
    """Two RRDBs under the pixel loss on patch shapes either side of the one-launch dense block's width limit (32), of the
    band height (2) and of the weight-gradient tile (4 x 32), odd sizes and the one-pixel image: every parameter gradient
    and the input gradient like for like with the oracle (<= 6 % per tensor; the full-depth net above measures 0.9 %)."""
    N, H, W = shape
    sd = O.center_outputs(O.make_generator_state(81, num_blocks=2), 0.5)
    gen = torch.Generator().manual_seed(82 + H * 64 + W)
    x0 = torch.rand(N, 3, H, W, generator=gen)
    target = torch.rand(N, 3, 4 * H, 4 * W, generator=gen)
    net = S.GeneratorRRDB(num_blocks=2)
    net.load_state_dict(sd)
    net = net.to(DEV).train()
    x = x0.to(DEV).requires_grad_(True)
    torch.nn.functional.l1_loss(net(x), target.to(DEV)).backward()
    with torch.no_grad():
        out_ref = O.generator_forward(sd, x0, emulate_bf16="hr_split")
    r = torch.sign(out_ref - target) / out_ref.numel()
    _o, ref_grads, ref_gx = O.generator_grads(sd, x0, r, input_grad=True, emulate_bf16="hr_split")
    (worst, name), errs = worst_error(((k, p.grad) for k, p in net.named_parameters()), ref_grads)
    assert len(errs) == 72 and worst < 6e-2, (worst, name)
    assert rel_l2(x.grad.cpu(), ref_gx) < 6e-2


def _with_fp64_accumulation(fn):
    """Runs ``fn`` with the oracle's convolutions accumulating in fp64 (same operands, same rounding points)."""
    import torch.nn.functional as F
    orig = F.conv2d

    def conv64(x, w, b=None, stride=1, padding=0):
        return orig(x.double(), w.double(), None if b is None else b.double(), stride=stride, padding=padding).float()
    F.conv2d = conv64
    try:
        return fn()
    finally:
        F.conv2d = orig


def test_two_block_reference_case_like_for_like(golden):
    """G3's inputs and white-noise cotangent, the case the fp32 comparison bounds at 20 %.  This net is chaotic at
    bf16: default-initialised pre-activations are tiny, a sum that lands on the other side of a bf16 rounding
    boundary moves the stored activation by a whole step, a few pre-activations downstream then take the other
    LeakyReLU slope, and an uncorrelated cotangent does not average that out.  The floor is measured here: the
    oracle against ITSELF with fp64 instead of fp32 accumulation (identical operands and rounding points) already
    differs by 1.6 % on the median tensor and 2.5 % on the worst.  The HIP path, whose summation order differs from
    both and whose backward also stores gradients in bf16, has to stay within 8 % on every tensor and 5 % on the
    median tensor (measured 7.3 % / 4.1 %; the smooth pixel-loss cotangent at full depth is held to 5 % above and
    measures 0.9 %)."""
    g = golden("g3_gen2.npz")
    sd = O.center_outputs(O.make_generator_state(int(g["seed"]), num_blocks=2), float(g["bias_shift"]))
    net = S.GeneratorRRDB(num_blocks=2)
    net.load_state_dict(sd)
    net = net.to(DEV).train()
    x = t(g["x"]).to(DEV).requires_grad_(True)
    net(x).backward(t(g["r"]).to(DEV))
    run = lambda: O.generator_grads(sd, t(g["x"]), t(g["r"]), input_grad=True, emulate_bf16="hr_split")
    _o, ref_grads, ref_gx = run()
    _o, ref64, _gx = _with_fp64_accumulation(run)
    floor = sorted(rel_l2(ref64[k], ref_grads[k]) for k in ref_grads)
    (worst, name), errs = worst_error(((k, p.grad) for k, p in net.named_parameters()), ref_grads)
    median = errs[len(errs) // 2][0]
    print("G3 like for like: worst", worst, name, " median", median, " | oracle fp32- vs fp64-accumulation: worst",
          floor[-1], " median", floor[len(floor) // 2])
    assert worst < 8e-2, (worst, name)
    assert median < 5e-2
    assert rel_l2(x.grad.cpu(), ref_gx) < 5e-2


def test_discriminator_gradients_like_for_like_away_from_initialisation():
    """A discriminator whose BatchNorm weights / biases / running buffers are perturbed (so that the normalisation's
    backward is not the near-cancellation of an untrained net), under the MSE-GAN loss, against autograd through the
    oracle's forward with bf16 conv operands.

    Two bf16 forward passes of one network (the HIP kernels', the emulation's) differ in their last bits, and wherever
    a pre-activation lies within that distance of zero the two LeakyReLUs take different slopes: about 0.4 % of the
    activations, each changing its gradient by a factor of five, which is 4-8 % in relative L2 on every gradient below
    (tools/diag_dgrad2.py: the difference appears behind the top two BatchNorm layers, 64 and 256 samples per channel,
    and nowhere else).  That is a property of comparing two rounded forwards, not of the backward pass: with the
    slopes pinned to the HIP forward's (the oracle's activations multiply by 1 / 0.2 where the HIP path's did) every
    gradient agrees within 2 % (measured 0.2-0.6 %); the unpinned comparison is kept with the bound that noise needs."""
    from super_resolution_amd import conv_core as cc
    sd = O.make_discriminator_state(81)
    gen = torch.Generator().manual_seed(82)
    for k in list(sd):
        if k.endswith("running_mean"):
            ci = k.rsplit(".", 1)[0]
            sd[ci + ".weight"] = 1.0 + 0.5 * torch.randn(sd[k].shape, generator=gen)
            sd[ci + ".bias"] = 0.3 * torch.randn(sd[k].shape, generator=gen)
    img = torch.rand(4, 3, 64, 64, generator=gen)
    netD = S.Discriminator()
    netD.load_state_dict(sd)
    netD = netD.to(DEV).train()
    # the inputs of the convs = the activations in front of them: their signs are the slopes the HIP forward took
    acts, plain = [], cc._conv_forward

    def recording(x, *a, **k):
        acts.append(x)
        return plain(x, *a, **k)

    cc._conv_forward = recording
    try:
        xg = img.to(DEV).requires_grad_(True)
        pred = netD(xg)
    finally:
        cc._conv_forward = plain
    loss = torch.nn.functional.mse_loss(pred, torch.ones_like(pred))
    loss.backward()
    masks = [(cc.blocked_as_nchw(a).float().cpu() > 0) for a in acts[1:]]
    pre_bn = {f"model.{ci}.bias" for ci, _a, _b, _s, bi in O.discriminator_layout() if bi is not None}
    for pinned, bound in ((True, 2e-2), (False, 9e-2)):
        leaves = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone())
                  for k, v in sd.items()}
        xr = img.clone().requires_grad_(True)
        pred_r = O.discriminator_forward(leaves, xr, True, emulate_bf16=True, slope_masks=masks if pinned else None)
        loss_r = torch.nn.functional.mse_loss(pred_r, torch.ones_like(pred_r))
        loss_r.backward()
        assert abs(loss.item() - loss_r.item()) < 2e-2 * abs(loss_r.item())
        errs = []
        for k, p in netD.named_parameters():
            if k in pre_bn:   # exactly zero in exact arithmetic (the normalisation removes a constant): noise on both sides
                continue
            errs.append((rel_l2(p.grad.cpu(), leaves[k].grad), k))
        errs.sort(reverse=True)
        e_in = rel_l2(xg.grad.cpu(), xr.grad)
        print(f"discriminator like for like, slopes {'pinned to the HIP forward' if pinned else 'free'}:",
              [(round(e, 4), k) for e, k in errs[:3]], " median", round(errs[len(errs) // 2][0], 4), " input", round(e_in, 4))
        assert errs[0][0] < bound and e_in < bound, (pinned, errs[0], e_in)


def _block_case(golden, name, module, specs):
    g = golden(name)
    gen = torch.Generator().manual_seed(int(g["seed"]))
    sd = O.make_conv_state(specs, gen)
    net = module.to(DEV).train()
    net.load_state_dict(sd)
    return g, sd, net


def _check_block_grads(g, sd, net, forward_fn, prefix_of):
    x = t(g["x"]).to(DEV).requires_grad_(True)
    out = net(x)
    assert out.requires_grad
    assert (out.detach().cpu() - t(g["out"])).abs().max().item() < 3e-2
    out.backward(t(g["r"]).to(DEV))
    # like for like: the oracle with bf16 conv operands, exact backward
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr = t(g["x"]).clone().requires_grad_(True)
    forward_fn(leaves, "", xr, emulate_bf16=True).backward(t(g["r"]))
    for k, p in net.named_parameters():
        assert rel_l2(p.grad.cpu(), leaves[k].grad) < 5e-2, k
    assert rel_l2(x.grad.cpu(), xr.grad) < 5e-2
    # and the reference's own fp32 gradients (white-noise cotangent on tiny pre-activations: the loose bound)
    assert rel_l2(x.grad.cpu(), t(g["grad_x"])) < 0.2
    for k, p in net.named_parameters():
        if f"gfull/{k}" in g:
            assert rel_l2(p.grad.cpu(), t(g[f"gfull/{k}"])) < 0.2, k
        got, ref = p.grad.double().abs().sum().item(), float(g[f"gabs/{k}"])
        assert abs(got - ref) < 0.15 * ref, (k, got, ref)


def test_standalone_dense_block_trains_g1(golden):
    specs = [(f"conv{k + 1}", 64 + 32 * k, 32 if k < 4 else 64) for k in range(5)]
    g, sd, net = _block_case(golden, "g1_rdb.npz", S.ResidualDenseBlock(64, 32), specs)
    _check_block_grads(g, sd, net, O.dense_block_forward, "")


def test_standalone_rrdb_trains_g2(golden):
    specs = [(f"rdb{r}.conv{k + 1}", 64 + 32 * k, 32 if k < 4 else 64) for r in (1, 2, 3) for k in range(5)]
    g, sd, net = _block_case(golden, "g2_rrdb.npz", S.RRDB(64, 32), specs)
    _check_block_grads(g, sd, net, O.rrdb_forward, "")


def test_standalone_blocks_train_on_images_wider_than_the_fused_kernel():
    """Width 40 > 32: the five-launch data-gradient path of the stand-alone block."""
    gen = torch.Generator().manual_seed(91)
    specs = [(f"conv{k + 1}", 64 + 32 * k, 32 if k < 4 else 64) for k in range(5)]
    sd = O.make_conv_state(specs, gen)
    net = S.ResidualDenseBlock(64, 32).to(DEV).train()
    net.load_state_dict(sd)
    x0, r = torch.randn(1, 64, 6, 40, generator=gen), torch.randn(1, 64, 6, 40, generator=gen)
    x = x0.to(DEV).requires_grad_(True)
    net(x).backward(r.to(DEV))
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr = x0.clone().requires_grad_(True)
    O.dense_block_forward(leaves, "", xr, emulate_bf16=True).backward(r)
    for k, p in net.named_parameters():
        assert rel_l2(p.grad.cpu(), leaves[k].grad) < 5e-2, k
    assert rel_l2(x.grad.cpu(), xr.grad) < 5e-2


def test_grad_scaler_steps_flat_adam():
    """train.py:161-163: scaler.scale(loss).backward(); scaler.step(optim); scaler.update() with FlatAdam in the
    optimizer's place: the unscaled update equals torch.optim.Adam's on the same gradients, and an overflowing step
    is skipped."""
    sd = O.make_generator_state(95, num_blocks=1)
    gen = torch.Generator().manual_seed(96)
    lr, hr = torch.rand(2, 3, 8, 8, generator=gen).to(DEV), torch.rand(2, 3, 32, 32, generator=gen).to(DEV)
    nets, opts = [], []
    for cls in (S.FlatAdam, torch.optim.Adam):
        net = S.GeneratorRRDB(num_blocks=1)
        net.load_state_dict(sd)
        net = net.to(DEV).train()
        nets.append(net)
        opts.append(cls(net.parameters(), lr=1e-3, betas=(0.9, 0.999)))
    scalers = [torch.amp.GradScaler("cuda", init_scale=1024.0) for _ in nets]
    for _ in range(2):
        for net, opt, sc in zip(nets, opts, scalers):
            opt.zero_grad()
            loss = torch.nn.functional.l1_loss(net(lr), hr)
            sc.scale(loss).backward()
            sc.step(opt)
            sc.update()
    for (k, a), (_k, b) in zip(nets[0].named_parameters(), nets[1].named_parameters()):
        assert not torch.equal(a.detach().cpu(), sd[k]), k                       # it moved
        assert torch.allclose(a.detach(), b.detach(), rtol=1e-4, atol=5e-5), k     # as torch.optim.Adam moves it
    # an infinite gradient: the step is skipped and the scale backs off
    net, opt, sc = nets[0], opts[0], scalers[0]
    before = {k: p.detach().clone() for k, p in net.named_parameters()}
    opt.zero_grad()
    loss = torch.nn.functional.l1_loss(net(lr), hr)
    sc.scale(loss).backward()
    next(net.parameters()).grad.view(-1)[0] = float("inf")
    scale_before = sc.get_scale()
    sc.step(opt)
    sc.update()
    assert sc.get_scale() < scale_before
    for k, p in net.named_parameters():
        assert torch.equal(p.detach(), before[k]), k


def test_adversarial_iteration_at_the_training_shape_is_finite_and_deterministic():
    """16 LR patches of 32 x 32, HR 128 x 128 (the shape train.py's defaults give): two runs from the same state give
    bit-identical losses and parameters, everything finite, every parameter of G and D moved."""
    sd_g = O.make_generator_state(101)
    sd_d = O.make_discriminator_state(102)
    sd_v = O.make_vgg_state(103)
    gen = torch.Generator().manual_seed(104)
    lr, hr = torch.rand(16, 3, 32, 32, generator=gen).to(DEV), torch.rand(16, 3, 128, 128, generator=gen).to(DEV)
    runs = []
    for _ in range(2):
        netG, netD, vgg = S.GeneratorRRDB(), S.Discriminator(), S.VGGFeatureExtractor(pretrained=False)
        netG.load_state_dict(sd_g), netD.load_state_dict(sd_d), vgg.load_state_dict(sd_v)
        netG, netD, vgg = netG.to(DEV).train(), netD.to(DEV).train(), vgg.to(DEV)
        optim_G = S.FlatAdam(netG.parameters(), lr=1e-4, betas=(0.9, 0.999))
        optim_D = S.FlatAdam(netD.parameters(), lr=1e-4, betas=(0.9, 0.999))
        out = [S.adversarial_step(netG, netD, vgg, optim_G, optim_D, lr, hr) for _ in range(2)][-1]
        torch.cuda.synchronize()
        runs.append((out, {k: p.detach().clone() for k, p in netG.named_parameters()},
                     {k: p.detach().clone() for k, p in netD.named_parameters()}))
    for k, v in runs[0][0].items():
        assert torch.isfinite(v).all(), k
        assert torch.equal(v, runs[1][0][k]), k
    for which in (1, 2):
        for k, p in runs[0][which].items():
            assert torch.isfinite(p).all(), k
            assert torch.equal(p, runs[1][which][k]), k
    assert all(not torch.equal(p.cpu(), sd_g[k]) for k, p in runs[0][1].items())
    assert all(not torch.equal(p.cpu(), sd_d[k]) for k, p in runs[0][2].items())


def test_flat_adam_takes_a_late_parameter_group_and_notices_parameters_that_left_its_buffer():
    torch.manual_seed(0)
    a, b = torch.nn.Linear(8, 8).to(DEV), torch.nn.Linear(8, 4).to(DEV)
    ra, rb = torch.nn.Linear(8, 8).to(DEV), torch.nn.Linear(8, 4).to(DEV)
    ra.load_state_dict(a.state_dict()), rb.load_state_dict(b.state_dict())
    opt, ref = S.FlatAdam(a.parameters(), lr=1e-2), torch.optim.Adam(ra.parameters(), lr=1e-2)
    opt.add_param_group({"params": list(b.parameters()), "lr": 1e-3})      # e.g. a layer unfrozen for fine-tuning
    ref.add_param_group({"params": list(rb.parameters()), "lr": 1e-3})
    x = torch.randn(5, 8, device=DEV)
    for _ in range(3):
        for (m1, m2, o) in ((a, b, opt), (ra, rb, ref)):
            o.zero_grad()
            m2(torch.tanh(m1(x))).square().mean().backward()
            o.step()
    for p, q in zip(list(a.parameters()) + list(b.parameters()), list(ra.parameters()) + list(rb.parameters())):
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-6)
    # a cast re-homes the parameters: the next step must not update a buffer the module no longer reads
    a.double().float()
    opt.zero_grad()
    b(torch.tanh(a(x))).square().mean().backward()
    with pytest.raises(RuntimeError, match="no longer aliases"):
        opt.step()
