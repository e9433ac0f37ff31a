"""GPU: what the drop-in modules do with the inputs a PyTorch caller can hand the reference's
(/root/reference/esrgan/models/esrgan_model.py:55-62, :86-87, :102-103): views that are not contiguous, other floating
dtypes, a batch of no images, and the inputs nn.Conv2d / nn.BatchNorm2d themselves refuse."""
import pytest
import torch

from oracle import esrgan_oracle as O
import super_resolution_amd as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def net():
    sd = O.center_outputs(O.make_generator_state(5, num_blocks=1))
    m = S.GeneratorRRDB(num_blocks=1)
    m.load_state_dict(sd)
    return m.to(DEV).eval()


def test_generator_reads_any_memory_format(net):
    x = torch.rand(2, 3, 12, 20, device=DEV)
    with torch.no_grad():
        ref = net(x)
        assert torch.equal(net(x.contiguous(memory_format=torch.channels_last)), ref)
        big = torch.rand(2, 3, 24, 40, device=DEV)
        big[:, :, ::2, 1::2] = x
        assert torch.equal(net(big[:, :, ::2, 1::2]), ref)
        assert torch.equal(net(x[:1].expand(2, 3, 12, 20)), net(x[:1].repeat(2, 1, 1, 1)))
        assert torch.equal(net(x.double()), ref)                         # every floating dtype is read as fp32
        assert (net(x.half()) - ref).abs().max().item() < 2e-3           # (the input's own rounding)
    # recorded forward and backward through a channels_last leaf
    net.train()
    try:
        a = x.clone().requires_grad_(True)
        b = x.contiguous(memory_format=torch.channels_last).clone().requires_grad_(True)
        net(a).square().sum().backward()
        ga = [p.grad.clone() for p in net.parameters()]
        net.zero_grad()
        net(b).square().sum().backward()
        assert torch.equal(a.grad, b.grad)
        assert all(torch.equal(g, p.grad) for g, p in zip(ga, net.parameters()))
    finally:
        net.zero_grad()
        net.eval()


def test_generator_refuses_what_conv2d_refuses(net):
    with pytest.raises(RuntimeError, match="should be the same"):
        net(torch.zeros(1, 3, 8, 8, dtype=torch.uint8, device=DEV))
    with pytest.raises(RuntimeError, match="NCHW"):
        net(torch.zeros(3, 8, 8, device=DEV))
    with pytest.raises(RuntimeError, match="NCHW"):
        net(torch.zeros(1, 4, 8, 8, device=DEV))
    with pytest.raises(RuntimeError, match="Kernel size"):
        net(torch.zeros(1, 3, 0, 8, device=DEV))


@pytest.mark.parametrize("cls", ["generator", "rrdb", "block"])
def test_a_batch_of_no_images_gives_no_images(cls, net):
    """nn.Conv2d maps [0, C, H, W] to [0, C', H, W]; a recorded call leaves all-zero gradients."""
    if cls == "generator":
        m, x, shape = net, torch.zeros(0, 3, 12, 20, device=DEV), (0, 3, 48, 80)
    elif cls == "rrdb":
        m, x, shape = S.RRDB(64, 32).to(DEV), torch.zeros(0, 64, 6, 5, device=DEV), (0, 64, 6, 5)
    else:
        m, x, shape = S.ResidualDenseBlock(64, 32).to(DEV), torch.zeros(0, 64, 6, 5, device=DEV), (0, 64, 6, 5)
    with torch.no_grad():
        out = m(x)
    assert out.shape == shape and out.dtype == torch.float32 and out.device == x.device
    m.train()
    try:
        xg = x.clone().requires_grad_(True)
        out = m(xg)
        assert out.shape == shape and out.requires_grad
        out.sum().backward()
        assert xg.grad is not None and xg.grad.shape == x.shape
        assert all(p.grad is not None and not bool(p.grad.any()) for p in m.parameters())
    finally:
        m.zero_grad()
        m.eval()


def test_discriminator_and_vgg_input_contract():
    D = S.Discriminator().to(DEV)
    V = S.VGGFeatureExtractor(pretrained=False).to(DEV).eval()
    x = torch.rand(2, 3, 48, 40, device=DEV)
    with torch.no_grad():
        assert torch.equal(V(x.contiguous(memory_format=torch.channels_last)), V(x))
        assert V(x[:0]).shape == (0, 512, 3, 2)
        assert V(torch.rand(1, 3, 50, 70, device=DEV)).shape == (1, 512, 3, 4)      # MaxPool2d floors odd sizes
        D.eval()
        assert torch.equal(D(x.contiguous(memory_format=torch.channels_last)), D(x))
        assert D(x[:0]).shape == (0, 1, 3, 3)
        assert D(torch.rand(2, 3, 50, 41, device=DEV)).shape == (2, 1, 4, 3)        # stride 2 / pad 1 rounds up
        assert D(torch.rand(1, 3, 16, 16, device=DEV)).shape == (1, 1, 1, 1)        # running statistics: any size
    for m in (D, V):
        with pytest.raises(RuntimeError, match="should be the same"):
            m(torch.zeros(1, 3, 16, 16, dtype=torch.uint8, device=DEV))
        with pytest.raises(RuntimeError, match="NCHW"):
            m(torch.zeros(3, 16, 16, device=DEV))
    D.train()
    with pytest.raises(ValueError, match="more than 1 value per channel"):          # nn.BatchNorm2d's own refusal
        D(torch.rand(1, 3, 16, 16, device=DEV))
    with pytest.raises(ValueError, match="more than 1 value per channel"):
        D.forward_pair(torch.rand(1, 3, 16, 16, device=DEV), torch.rand(1, 3, 16, 16, device=DEV))
    assert D(torch.rand(2, 3, 16, 16, device=DEV)).shape == (2, 1, 1, 1)


def test_module_level_operations(net):
    """deepcopy, pickling, a round trip over the CPU, casts and partial freezing, as nn.Module users apply them."""
    import copy
    import io
    x = torch.rand(2, 3, 12, 20, device=DEV)
    with torch.no_grad():
        ref = net(x)
        assert torch.equal(copy.deepcopy(net)(x), ref)
        buf = io.BytesIO()
        torch.save(net, buf)
        buf.seek(0)
        assert torch.equal(torch.load(buf, weights_only=False)(x), ref)      # (a file this test wrote itself)
        moved = copy.deepcopy(net).cpu()
        with pytest.raises(RuntimeError, match="ROCm"):
            moved(x.cpu())
        assert torch.equal(moved.to(DEV)(x), ref)
        # a cast module computes as before (bf16 kernels on fp32 copies of its parameters) and answers in its dtype
        for cast, tol in ((torch.float16, 2e-3), (torch.bfloat16, 2e-2), (torch.float64, 0.0)):
            m = copy.deepcopy(net).to(cast)
            out = m(x.to(cast))
            assert out.dtype == cast
            assert (out.float() - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item()), cast
            blk = copy.deepcopy(net.body[0]).to(cast)
            assert blk(torch.randn(1, 64, 6, 7, device=DEV, dtype=cast)).dtype == cast
    with pytest.raises(RuntimeError, match="fp32 parameters"):
        copy.deepcopy(net).half().train()(x.half())
    # frozen parameters get no gradient, the others the same gradients as without the freeze
    a, b = copy.deepcopy(net).train(), copy.deepcopy(net).train()
    a.conv_first.requires_grad_(False)
    a.body[0].rdb2.conv3.weight.requires_grad_(False)
    a(x).square().mean().backward()
    b(x).square().mean().backward()
    frozen = [p.grad is None for p in a.parameters()]
    assert sum(frozen) == 3
    assert all(torch.equal(p.grad, q.grad) for p, q in zip(a.parameters(), b.parameters()) if p.grad is not None)


def test_second_order_gradients_are_refused_not_wrong(net):
    """The backward kernels are not differentiable a second time: asking for it raises instead of returning a graph that
    ignores them."""
    import copy
    m = copy.deepcopy(net).train()
    x = torch.rand(1, 3, 8, 8, device=DEV, requires_grad=True)
    g, = torch.autograd.grad(m(x).square().mean(), x, create_graph=True)
    with pytest.raises(RuntimeError, match="once_differentiable|differentiated twice|twice"):
        g.square().sum().backward()


def test_byte_edge_and_tiling_on_unusual_inputs(net):
    from super_resolution_amd import sharded
    img = (torch.rand(13, 17, 3, device=DEV) * 255).to(torch.uint8)
    ref = net.upscale_u8(img)
    assert torch.equal(net.upscale_u8(img.permute(2, 0, 1).contiguous().permute(1, 2, 0)), ref)    # a strided HWC view
    assert torch.equal(net.upscale_u8(img[None].repeat(3, 1, 1, 1))[2], ref)
    assert net.upscale_u8(img[:1, :1]).shape == (4, 4, 3)
    assert net.upscale_u8(img[None][:0]).shape == (0, 52, 68, 3)
    for bad in (img.float(), img.permute(2, 0, 1), img[:0]):
        with pytest.raises(RuntimeError):
            net.upscale_u8(bad)
    exact = S.GeneratorRRDB(num_blocks=1, precision="fp32")
    exact.load_state_dict(net.state_dict())
    exact = exact.to(DEV).eval()
    with torch.no_grad():
        for shape, tiles in (((1, 3, 5, 7), 4), ((1, 3, 40, 33), 1), ((2, 3, 40, 33), 6), ((1, 3, 40, 33), 64)):
            x = torch.rand(*shape, device=DEV)
            # (more tiles than the halo leaves room for: every window is the whole image)
            assert (sharded.tiled_forward(exact, x, tiles) - exact(x)).abs().max().item() < 1e-5, (shape, tiles)
        for tiles, halo in ((0, 8), (2, -1)):
            with pytest.raises(ValueError):
                sharded.tiled_forward(exact, torch.rand(1, 3, 8, 8, device=DEV), tiles, halo)
        with pytest.raises(ValueError, match="cannot be cut"):
            sharded.tiled_forward(exact, torch.rand(1, 3, 1, 1, device=DEV), 4)


def test_widened_models_refuse_what_their_first_conv_refuses():
    from super_resolution_amd import cvae, cyclegan
    G, Dn, M = cyclegan.GeneratorSR().to(DEV).eval(), cyclegan.GeneratorDown().to(DEV).eval(), cvae.CVAE().to(DEV).eval()
    bytes_ = torch.zeros(1, 3, 32, 32, dtype=torch.uint8, device=DEV)
    with torch.no_grad():
        for m in (G, Dn, M):
            with pytest.raises(RuntimeError, match="should be the same"):
                m(bytes_)
        assert G(torch.rand(2, 3, 9, 11, device=DEV)).shape == (2, 3, 36, 44)
        assert Dn(torch.rand(1, 3, 37, 45, device=DEV)).shape == (1, 3, 9, 11)
        with pytest.raises(RuntimeError, match="empty batch"):
            G(torch.rand(0, 3, 9, 11, device=DEV))
