"""CPU (gloo, world_size 2 and 3): the tile grid / crop / gather logic of tile-sharded inference, with the
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


def test_tile_grid_partitions_the_image_with_windows_of_one_size():
    for H, W, ty, tx, halo in ((339, 510, 2, 4, 8), (339, 510, 8, 1, 16), (10, 7, 4, 3, 3), (5, 5, 8, 8, 2), (1, 1, 1, 1, 0)):
        tiles = sharded.tile_edges(H, W, ty, tx, halo)
        owned = torch.zeros(H, W, dtype=torch.int32)
        sizes = set()
        for y0, y1, x0, x1, a, b, c, d in tiles:
            owned[y0:y1, x0:x1] += 1
            assert 0 <= a <= y0 < y1 <= b <= H and 0 <= c <= x0 < x1 <= d <= W
            # the window holds the tile's halo as far as the image goes
            assert a <= max(0, y0 - halo) and b >= min(H, y1 + halo) and c <= max(0, x0 - halo) and d >= min(W, x1 + halo)
            sizes.add((b - a, d - c))
        assert bool((owned == 1).all()) and len(sizes) == 1
    with pytest.raises(ValueError):
        sharded.tile_edges(0, 4, 1, 1, 1)


def test_grid_choice_minimises_the_pixels_computed():
    # BASELINE config 4: a 510 x 339 LR image on 8 ranks
    assert sharded.best_grid(339, 510, 8, 8) == (2, 4)
    f_grid = sharded.work_factor(339, 510, sharded.tile_edges(339, 510, 2, 4, 8))
    f_rows = sharded.work_factor(339, 510, sharded.tile_edges(339, 510, 8, 1, 8))
    f_rows16 = sharded.work_factor(339, 510, sharded.tile_edges(339, 510, 8, 1, 16))
    assert f_grid < 1.2 < f_rows < 1.45 and 1.7 < f_rows16 < 1.8
    assert sharded.best_grid(100, 20, 4, 8) == (4, 1) and sharded.best_grid(20, 100, 4, 8) == (1, 4)
    assert sharded.best_grid(64, 64, 1, 8) == (1, 1)
    with pytest.raises(ValueError):
        sharded.best_grid(2, 2, 7, 1)


def test_tiled_forward_against_the_whole_image():
    sd = O.make_generator_state(7, num_blocks=2)
    x = torch.rand(1, 3, 30, 18, generator=torch.Generator().manual_seed(8))
    net = lambda t: O.generator_forward(sd, t)
    with torch.no_grad():
        full = O.generator_forward(sd, x)
        # a halo as large as the image: every window is the whole image, the result is the whole-image result
        for grid in ((3, 1), (2, 2), None):
            assert (sharded.tiled_forward(net, x, bands=4 if grid != (3, 1) else 3, halo=30, grid=grid) - full).abs().max().item() < 1e-5
        # the error falls with the halo, whatever the cut
        errs = [(sharded.tiled_forward(net, x, 4, halo=h, grid=(2, 2)) - full).abs().max().item() for h in (1, 3, 6)]
        assert errs[0] > errs[1] > errs[2] and errs[2] < 1e-3
        # one batched forward over the tiles = the tiles one after the other
        a = sharded.tiled_forward(net, x, 4, halo=3, batched=True)
        b = sharded.tiled_forward(net, x, 4, halo=3, batched=False)
        assert a.shape == full.shape and (a - b).abs().max().item() < 1e-5
        # row bands as before
        assert sharded.tiled_forward(net, x, bands=3, halo=5, grid=(3, 1)).shape == full.shape


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
            ref = sharded.tiled_forward(net, x, world, halo=4, batched=False)      # the same tiles on one rank
            ret["equal"] = bool(torch.equal(out, ref))
            full = O.generator_forward(sd, x)
            ret["close"] = (out - full).abs().max().item()
            ret["shape"] = tuple(out.shape)
        else:
            assert out is None
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_forward_gloo(world):
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert ret["equal"] and ret["shape"] == (2, 3, 84, 48) and ret["close"] < 5e-2
