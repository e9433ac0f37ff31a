"""Tile sharding of one large image, across ranks (one process per GPU) or as one batch on one GPU.

The reference runs the whole LR image through the generator in one call
(/root/reference/esrgan/test.py:71-72).  Here the image is cut into a ``ty x tx`` grid of tiles, each
extended by ``halo`` LR pixels on its interior edges; every tile is super-resolved on its own, the
``4*halo`` HR pixels that only served as context are cropped, and the tiles are put together.  There
is no collective inside the network: across ranks the only exchange is the final gather.

* The grid is chosen to minimise the pixels computed: a 510x339 image on 8 ranks is 2 x 4 tiles
  (1.18 x the image's pixels at halo 8) rather than 8 row bands (1.38 x; 1.76 x at halo 16).
* Every tile is computed on a window of one common size (a tile at the image border takes its extra
  context from the interior), so the tiles of one GPU run as ONE batched forward: the batch fills the
  chip where eight small forwards, one launch per conv each, would not.

The generator's receptive field is 347 stacked 3x3 convs, wider than any practical halo, so the
result is an approximation whose error falls with ``halo``.  With default-initialised weights (the
only ones there are: the reference ships no ESRGAN checkpoint): halo 4 -> 8e-3 max-abs, halo 8 ->
5e-6, halo 16 -> 1e-6 against the whole-image forward in fp32.  The default, 8, is the smallest that
meets the 1e-3 bound of the parity gate (tests/test_gpu_parity.py); trained weights have a wider
effective field: raise ``halo`` for them.  ``halo >= max(H, W)`` reproduces the full-image result
exactly.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch
from torch import Tensor

DEFAULT_HALO = 8

# (y0, y1, x0, x1, a, b, c, d): rows [y0, y1) x columns [x0, x1) are owned, rows [a, b) x columns [c, d) are computed
Tile = Tuple[int, int, int, int, int, int, int, int]


def _cuts(length: int, parts: int) -> List[int]:
    return [round(i * length / parts) for i in range(parts + 1)]


def _window(lo: int, hi: int, size: int, length: int, halo: int) -> Tuple[int, int]:
    """A window of ``size`` pixels inside [0, length) that holds [lo - halo, hi + halo) as far as the image goes."""
    a = min(max(lo - halo, 0), length - size)
    return a, a + size


def tile_edges(height: int, width: int, tiles_y: int, tiles_x: int, halo: int) -> List[Tile]:
    """The tiles of a ``tiles_y x tiles_x`` grid, row-major; all computed windows have one size."""
    if height < 1 or width < 1 or tiles_y < 1 or tiles_x < 1 or halo < 0:
        raise ValueError("image size and tile counts must be positive, halo non-negative")
    tiles_y, tiles_x = min(tiles_y, height), min(tiles_x, width)
    ys, xs = _cuts(height, tiles_y), _cuts(width, tiles_x)
    # context on both sides only where a tile has two interior edges (three or more tiles along the axis)
    hc = min(height, max(ys[i + 1] - ys[i] for i in range(tiles_y)) + min(tiles_y - 1, 2) * halo)
    wc = min(width, max(xs[j + 1] - xs[j] for j in range(tiles_x)) + min(tiles_x - 1, 2) * halo)
    out = []
    for i in range(tiles_y):
        a, b = _window(ys[i], ys[i + 1], hc, height, halo)
        for j in range(tiles_x):
            c, d = _window(xs[j], xs[j + 1], wc, width, halo)
            out.append((ys[i], ys[i + 1], xs[j], xs[j + 1], a, b, c, d))
    return out


def work_factor(height: int, width: int, tiles: List[Tile]) -> float:
    """Pixels computed over pixels of the image."""
    return sum((t[5] - t[4]) * (t[7] - t[6]) for t in tiles) / float(height * width)


def best_grid(height: int, width: int, n: int, halo: int) -> Tuple[int, int]:
    """(tiles_y, tiles_x) with tiles_y * tiles_x == n that computes the fewest pixels."""
    if n < 1:
        raise ValueError("the number of tiles must be positive")
    best = None
    for ty in range(1, n + 1):
        if n % ty or ty > height or n // ty > width:
            continue
        f = work_factor(height, width, tile_edges(height, width, ty, n // ty, halo))
        if best is None or f < best[0] - 1e-12:
            best = (f, ty, n // ty)
    if best is None:
        raise ValueError(f"a {height}x{width} image cannot be cut into {n} tiles")
    return best[1], best[2]


def band_edges(height: int, bands: int, halo: int) -> List[Tuple[int, int, int, int]]:
    """(y0, y1, a, b) per row band: rows [y0, y1) are owned, rows [a, b) are computed (bands of their own sizes)."""
    if height < 1 or bands < 1 or halo < 0:
        raise ValueError("height and bands must be positive, halo non-negative")
    bands = min(bands, height)
    cuts = _cuts(height, bands)
    return [(cuts[i], cuts[i + 1], max(0, cuts[i] - halo), min(height, cuts[i + 1] + halo))
            for i in range(bands)]


def _crop(sr: Tensor, t: Tile, scale: int) -> Tensor:
    y0, y1, x0, x1, a, _b, c, _d = t
    return sr[..., scale * (y0 - a): scale * (y1 - a), scale * (x0 - c): scale * (x1 - c)]


def run_tile(net: Callable[[Tensor], Tensor], x: Tensor, t: Tile, scale: int = 4) -> Tensor:
    """Super-resolve the computed window of tile ``t`` and return the HR pixels it owns."""
    return _crop(net(x[:, :, t[4]:t[5], t[6]:t[7]].contiguous()), t, scale)


def _assemble(parts: List[Tensor], tiles_y: int, tiles_x: int) -> Tensor:
    rows = [torch.cat(parts[i * tiles_x:(i + 1) * tiles_x], dim=3) for i in range(tiles_y)]
    return torch.cat(rows, dim=2)


def tiled_forward(net: Callable[[Tensor], Tensor], x: Tensor, bands: int, halo: int = DEFAULT_HALO, scale: int = 4,
                  grid: Optional[Tuple[int, int]] = None, batched: bool = True) -> Tensor:
    """``bands`` tiles on the calling device.  ``grid`` = (tiles_y, tiles_x) fixes the cut ((bands, 1): row bands);
    by default the grid that computes the fewest pixels.  ``batched``: all tiles as one batch through one forward
    (fast: the batch fills the GPU); ``False`` runs them one after the other, which bounds the activation memory to
    one tile's."""
    H, W = int(x.shape[2]), int(x.shape[3])
    ty, tx = grid if grid is not None else best_grid(H, W, bands, halo)
    tiles = tile_edges(H, W, ty, tx, halo)
    ty, tx = min(ty, H), min(tx, W)
    if batched:
        N = x.shape[0]
        stack = torch.cat([x[:, :, t[4]:t[5], t[6]:t[7]] for t in tiles], dim=0)          # [tiles * N, C, hc, wc]
        sr = net(stack.contiguous())
        parts = [_crop(sr[k * N:(k + 1) * N], t, scale) for k, t in enumerate(tiles)]
    else:
        parts = [run_tile(net, x, t, scale) for t in tiles]
    return _assemble(parts, ty, tx)


def sharded_forward(net: Callable[[Tensor], Tensor], x: Tensor, halo: int = DEFAULT_HALO, scale: int = 4,
                    group=None, dst: int = 0, out_channels: int = 3,
                    grid: Optional[Tuple[int, int]] = None) -> Optional[Tensor]:
    """Each rank computes tile ``rank`` of ``x`` (every rank holds the same LR image; the grid has one tile per rank,
    by default the one that computes the fewest pixels); rank ``dst`` returns the assembled HR image, the others
    ``None``.  Works with any backend (RCCL on GPUs)."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    N, H, W = int(x.shape[0]), int(x.shape[2]), int(x.shape[3])
    if grid is None:
        n = world
        while n > 1 and n > H * W:
            n -= 1
        # a world size the image cannot be cut into evenly leaves the last ranks idle
        while True:
            try:
                ty, tx = best_grid(H, W, n, halo)
                break
            except ValueError:
                n -= 1
    else:
        ty, tx = grid
    tiles = tile_edges(H, W, ty, tx, halo)
    ty, tx = min(ty, H), min(tx, W)
    if len(tiles) > world:
        raise ValueError(f"a {ty}x{tx} grid needs {len(tiles)} ranks, the group has {world}")
    rows = max(scale * (t[1] - t[0]) for t in tiles)
    cols = max(scale * (t[3] - t[2]) for t in tiles)
    # gather needs equal shapes: pad every tile to the largest one (ranks beyond the tile count send zeros)
    padded = torch.zeros((N, out_channels, rows, cols), dtype=torch.float32, device=x.device)
    if rank < len(tiles):
        mine = run_tile(net, x, tiles[rank], scale)
        padded[:, :, : mine.shape[2], : mine.shape[3]] = mine
    # RCCL gathers device tensors; on gloo (tests, CPU-only fabrics) a device tensor is staged through host memory
    # (gloo's own device-tensor collectives stall when ranks share a GPU)
    staged = padded.is_cuda and dist.get_backend(group) != "nccl"
    send = padded.cpu() if staged else padded
    bucket = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, bucket, dst=dst, group=group)
    if rank != dst:
        return None
    parts = [bucket[i][:, :, : scale * (t[1] - t[0]), : scale * (t[3] - t[2])] for i, t in enumerate(tiles)]
    out = _assemble(parts, ty, tx)
    return out.to(x.device) if staged else out
