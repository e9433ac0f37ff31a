"""CPU (gloo, world_size 2): the band split / crop / gather logic of tile-sharded inference, with the
oracle standing in for the network so the test needs no GPU."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import esrgan_oracle as O
from super_resolution_amd import sharded


def test_band_edges_cover_the_image_once():
    for H, bands, halo in ((339, 8, 16), (10, 4, 3), (5, 8, 2), (1, 1, 0)):
        edges = sharded.band_edges(H, bands, halo)
        assert edges[0][0] == 0 and edges[-1][1] == H
        for (y0, y1, a, b), nxt in zip(edges, edges[1:] + [None]):
            assert a <= y0 < y1 <= b and a == max(0, y0 - halo) and b == min(H, y1 + halo)
            if nxt:
                assert nxt[0] == y1
    with pytest.raises(ValueError):
        sharded.band_edges(0, 1, 1)


def test_tiled_forward_matches_oracle_restatement():
    sd = O.make_generator_state(7, num_blocks=2)
    x = torch.rand(1, 3, 30, 18, generator=torch.Generator().manual_seed(8))
    net = lambda t: O.generator_forward(sd, t)
    with torch.no_grad():
        assert torch.equal(sharded.tiled_forward(net, x, bands=3, halo=5), O.generator_forward_tiled(sd, x, 3, 5))
        full = O.generator_forward(sd, x)
        assert (sharded.tiled_forward(net, x, bands=3, halo=30) - full).abs().max().item() < 1e-5


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    sd = O.make_generator_state(7, num_blocks=1)
    x = torch.rand(2, 3, 21, 12, generator=torch.Generator().manual_seed(9))
    net = lambda t: O.generator_forward(sd, t)
    with torch.no_grad():
        out = sharded.sharded_forward(net, x, halo=4)
        if rank == 0:
            ref = O.generator_forward_tiled(sd, x, world, 4)
            ret["equal"] = bool(torch.equal(out, ref))
            ret["shape"] = tuple(out.shape)
        else:
            assert out is None
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_forward_gloo(world):
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert ret["equal"] and ret["shape"] == (2, 3, 84, 48)
