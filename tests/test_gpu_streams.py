"""GPU: the library launches on the caller's stream (torch's current stream of the device) and ties its own helper streams
to it with events.  Everything here runs under a non-default stream that is kept busy, with the default stream idle: a
launch or an event wait that went to the wrong stream would read inputs that are not there yet or return before its work
is done.  Results must be those of the default-stream run, bit for bit.  Call sites: /root/reference/esrgan/train.py:157-163,
:196-223 and test.py:71-72 run on PyTorch's current stream as well."""
import pytest
import torch

from oracle import esrgan_oracle as O
import super_resolution_amd as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def busy(stream, ms=3.0):
    """Keeps ``stream`` occupied for a few milliseconds, so that work queued behind it really is behind it."""
    a = torch.empty(4096, 4096, device=DEV)
    with torch.cuda.stream(stream):
        a.normal_()
        for _ in range(int(ms * 2)):
            a = a @ a * 1e-4
    return a


def on_stream(fn):
    side = torch.cuda.Stream(device=DEV)
    torch.cuda.synchronize()
    keep = busy(side)
    with torch.cuda.stream(side):
        out = fn()
    side.synchronize()
    del keep
    return out


@pytest.mark.parametrize("shape", [(2, 3, 24, 20), (4, 3, 128, 128)])
def test_inference_forward_on_a_side_stream(shape):
    """The second shape takes the persistent launch of the body (128 tiles) and the two-part HR side."""
    sd = O.center_outputs(O.make_generator_state(41, num_blocks=2))
    net = S.GeneratorRRDB(num_blocks=2)
    net.load_state_dict(sd)
    net = net.to(DEV).eval()
    x = torch.rand(*shape, device=DEV)
    with torch.no_grad():
        ref = net(x)
        torch.cuda.synchronize()
        out = on_stream(lambda: net(x))
        assert torch.equal(out, ref)
        img = (torch.rand(33, 47, 3, device=DEV) * 255).to(torch.uint8)
        ref8 = net.upscale_u8(img)
        torch.cuda.synchronize()
        assert torch.equal(on_stream(lambda: net.upscale_u8(img)), ref8)


@pytest.mark.parametrize("hw", [(32, 32), (40, 48)])
def test_training_iteration_on_a_side_stream(hw):
    """Pixel iteration (recorded forward, backward with its weight-gradient side stream where one is used, FlatAdam):
    parameters after two iterations equal those of the default-stream run."""
    sd = O.center_outputs(O.make_generator_state(43, num_blocks=2))
    lr = torch.rand(4, 3, *hw, device=DEV)
    hr = torch.rand(4, 3, 4 * hw[0], 4 * hw[1], device=DEV)

    def run(side):
        net = S.GeneratorRRDB(num_blocks=2)
        net.load_state_dict(sd)
        net = net.to(DEV).train()
        opt = S.FlatAdam(net.parameters(), lr=1e-3)
        torch.cuda.synchronize()

        def steps():
            for _ in range(2):
                S.pixel_step(net, opt, lr, hr)
        on_stream(steps) if side else steps()
        torch.cuda.synchronize()
        return [p.detach().clone() for p in net.parameters()]

    a, b = run(False), run(True)
    assert all(torch.equal(p, q) for p, q in zip(a, b))


def test_adversarial_iteration_on_a_side_stream():
    sd_g = O.center_outputs(O.make_generator_state(45, num_blocks=1))
    lr = torch.rand(2, 3, 16, 16, device=DEV)
    hr = torch.rand(2, 3, 64, 64, device=DEV)

    def run(side):
        torch.manual_seed(7)
        netG = S.GeneratorRRDB(num_blocks=1)
        netG.load_state_dict(sd_g)
        netG = netG.to(DEV).train()
        netD = S.Discriminator().to(DEV).train()
        vgg = S.VGGFeatureExtractor(pretrained=False).to(DEV).eval()
        og = S.FlatAdam(netG.parameters(), lr=1e-3)
        od = S.FlatAdam(netD.parameters(), lr=1e-3)
        torch.cuda.synchronize()

        def steps():
            for _ in range(2):
                S.adversarial_step(netG, netD, vgg, og, od, lr, hr)
        on_stream(steps) if side else steps()
        torch.cuda.synchronize()
        return [p.detach().clone() for m in (netG, netD) for p in list(m.parameters()) + list(m.buffers())]

    a, b = run(False), run(True)
    assert len(a) == len(b) and all(torch.equal(p, q) for p, q in zip(a, b))
