"""GPU: the body of the generator as one persistent launch (per-tile pass counters instead of kernel boundaries)
against one launch per conv (/root/reference/esrgan/models/esrgan_model.py:55-58).  With every pass walking its input
channels in channel order (SR355_CHAIN_PERM=0) both run the same stage loop on the same tiles in the same accumulation
order, so the outputs must be bit-identical; repeated runs must be too (a stale read of a neighbour's tile would show up
as run-to-run differences).  The default stage order (a pass starts with the channels the pass before it touched last,
for the XCD's L2) changes the summation order only: checked against the channel-order result within fp32 summation
noise, and bit for bit against itself."""
import os

import pytest
import torch

from oracle import esrgan_oracle as O
import super_resolution_amd as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def run(net, x, chain, perm=False):
    os.environ["SR355_CHAIN"] = "1" if chain else "0"
    os.environ["SR355_CHAIN_PERM"] = "1" if perm else "0"
    try:
        with torch.no_grad():
            y = net(x)
        torch.cuda.synchronize()
        return y
    finally:
        os.environ.pop("SR355_CHAIN", None)
        os.environ.pop("SR355_CHAIN_PERM", None)


@pytest.mark.parametrize("shape,blocks,kw", [
    ((8, 3, 128, 128), 2, {}),                                   # 256 tiles: one per workgroup on half the slots
    ((16, 3, 128, 128), 3, {"dense_skip_fp32": True}),           # the benchmark batch, an fp32 carrier for every skip
    ((3, 3, 200, 330), 2, {"hr_f16": False}),                    # ragged edges: 13 x 11 tiles per image
    ((40, 3, 128, 96), 1, {}),                                   # more tiles (960) than resident workgroups
    ((1, 3, 512, 256), 1, {}),                                   # one image, exactly as many tiles as the threshold
    ((1, 3, 16, 8192), 1, {}),                                   # one row of 256 tiles: left / right neighbours only
    ((1, 3, 4100, 32), 1, {}),                                   # one column of 257 tiles, the last one partial
])
def test_chain_is_bit_identical_to_one_launch_per_conv(shape, blocks, kw):
    sd = O.center_outputs(O.make_generator_state(7, num_blocks=blocks), 0.5)
    net = S.GeneratorRRDB(num_blocks=blocks, **kw)
    net.load_state_dict(sd)
    net = net.to(DEV).eval()
    x = torch.rand(*shape, generator=torch.Generator().manual_seed(8)).to(DEV)
    ref = run(net, x, chain=False)
    for _ in range(3):
        got = run(net, x, chain=True)
        assert torch.equal(got, ref)
    # the default stage order: another summation order of the same products
    first = run(net, x, chain=True, perm=True)
    for _ in range(2):
        assert torch.equal(run(net, x, chain=True, perm=True), first)
    err = (first.double() - ref.double()).pow(2).mean().sqrt().item()
    assert err < max(3e-4, 5e-3 * ref.double().std().item()), err     # bf16 rounding flips of another summation order


def test_full_depth_chain_matches_and_repeats():
    sd = O.center_outputs(O.make_generator_state(9), 0.5)
    net = S.GeneratorRRDB()
    net.load_state_dict(sd)
    net = net.to(DEV).eval()
    x = torch.rand(16, 3, 128, 128, generator=torch.Generator().manual_seed(10)).to(DEV)
    ref = run(net, x, chain=False)
    for _ in range(5):
        assert torch.equal(run(net, x, chain=True), ref)
    first = run(net, x, chain=True, perm=True)
    for _ in range(3):
        assert torch.equal(run(net, x, chain=True, perm=True), first)
    err = (first.double() - ref.double()).pow(2).mean().sqrt().item()
    print(f"stage-order permutation vs channel order: rms {err:.3e} (output std {ref.double().std().item():.3e})")
    assert err < 5e-3 * ref.double().std().item(), err     # (69 chained blocks amplify bf16 rounding flips: 2e-3 measured)
    # the image path shares the workspace and the chain
    img = torch.randint(0, 256, (16, 128, 128, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(11)).to(DEV)
    os.environ["SR355_CHAIN"] = "0"
    os.environ["SR355_CHAIN_PERM"] = "0"
    try:
        ref8 = net.upscale_u8(img)
        os.environ["SR355_CHAIN"] = "1"
        assert torch.equal(net.upscale_u8(img), ref8)
    finally:
        os.environ.pop("SR355_CHAIN", None)
        os.environ.pop("SR355_CHAIN_PERM", None)


def test_a_launch_that_gives_up_is_reported_on_the_same_call_path_and_retried(monkeypatch):
    """The body's persistent launch needs every workgroup resident.  Forced to a grid that cannot be (one workgroup per
    tile of an image with more tiles than slots, so that resident tiles border tiles that are not: SR355_CHAIN_GRID),
    its waits give up after two seconds.  The status call
    then reports that launch (once, and also when good launches were queued behind it in the meantime), the device falls
    back to one launch per conv, and a ``checked`` module returns the valid result from the same call."""
    import warnings
    from super_resolution_amd import ops
    from super_resolution_amd._lib import Sr355Error
    sd = O.make_generator_state(5, num_blocks=1)
    net = S.GeneratorRRDB(num_blocks=1)
    net.load_state_dict(sd)
    net = net.to(DEV).eval()
    x = torch.rand(1, 3, 512, 1280, generator=torch.Generator().manual_seed(6)).to(DEV)    # 32 x 40 tiles of 16 x 32
    with torch.no_grad():
        monkeypatch.setenv("SR355_CHAIN_PERM", "0")
        monkeypatch.setenv("SR355_CHAIN", "0")
        ref = net(x).clone()
        monkeypatch.delenv("SR355_CHAIN")
        good = net(x).clone()
        assert ops.chain_status(True) is True and torch.equal(good, ref)
        try:
            monkeypatch.setenv("SR355_CHAIN_GRID", "1280")
            net(x)
            with pytest.raises(Sr355Error, match="gave up"):
                ops.chain_status(True)
            assert ops.chain_status(True) is True                       # reported once
            monkeypatch.delenv("SR355_CHAIN_GRID")
            assert torch.equal(net(x), ref)                             # the device runs one launch per conv now
            ops.chain_reenable()
            # a failure is not forgotten when good launches are queued behind it before anyone asks
            monkeypatch.setenv("SR355_CHAIN_GRID", "1280")
            net(x)
            monkeypatch.delenv("SR355_CHAIN_GRID")
            later = net(x)
            with pytest.raises(Sr355Error, match="gave up"):
                ops.chain_status(True)
            assert torch.equal(later, ref) and ops.chain_status(True) is True
            ops.chain_reenable()
            monkeypatch.setenv("SR355_CHAIN_GRID", "1280")
            net.checked = True
            with warnings.catch_warnings(record=True) as seen:
                warnings.simplefilter("always")
                out = net(x)
            assert torch.equal(out, ref) and any("repeated as one launch per conv" in str(w.message) for w in seen)
        finally:
            monkeypatch.delenv("SR355_CHAIN_GRID", raising=False)
            net.checked = False
            ops.chain_reenable()
        assert torch.equal(net(x), ref) and ops.chain_status(True) is True
