"""GPU: the body of the generator as one persistent launch (per-tile pass counters instead of kernel boundaries)
against one launch per conv (/root/reference/esrgan/models/esrgan_model.py:55-58).  Both run the same stage loop on
the same tiles in the same accumulation order, so the outputs must be bit-identical; repeated runs must be too (a
stale read of a neighbour's tile would show up as run-to-run differences)."""
import os

import pytest
import torch

from oracle import esrgan_oracle as O
import super_resolution_amd as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def run(net, x, chain):
    os.environ["SR355_CHAIN"] = "1" if chain else "0"
    try:
        with torch.no_grad():
            y = net(x)
        torch.cuda.synchronize()
        return y
    finally:
        os.environ.pop("SR355_CHAIN", None)


@pytest.mark.parametrize("shape,blocks,kw", [
    ((8, 3, 128, 128), 2, {}),                                   # 256 tiles: one per workgroup on half the slots
    ((16, 3, 128, 128), 3, {"dense_skip_fp32": True}),           # the benchmark batch, an fp32 carrier for every skip
    ((3, 3, 200, 330), 2, {"hr_f16": False}),                    # ragged edges: 13 x 11 tiles per image
    ((40, 3, 128, 96), 1, {}),                                   # more tiles (960) than resident workgroups
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


def test_full_depth_chain_matches_and_repeats():
    sd = O.center_outputs(O.make_generator_state(9), 0.5)
    net = S.GeneratorRRDB()
    net.load_state_dict(sd)
    net = net.to(DEV).eval()
    x = torch.rand(16, 3, 128, 128, generator=torch.Generator().manual_seed(10)).to(DEV)
    ref = run(net, x, chain=False)
    for _ in range(5):
        assert torch.equal(run(net, x, chain=True), ref)
    # the image path shares the workspace and the chain
    img = torch.randint(0, 256, (16, 128, 128, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(11)).to(DEV)
    os.environ["SR355_CHAIN"] = "0"
    try:
        ref8 = net.upscale_u8(img)
    finally:
        os.environ.pop("SR355_CHAIN", None)
    assert torch.equal(net.upscale_u8(img), ref8)
