"""CPU (gloo, world_size 2 and 3): bucket planning and the overlap orchestration of the data-parallel backward,
with a stand-in for the HIP phase executor; the started-then-awaited mean the discriminator's optimizer uses."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import esrgan_oracle as O
from super_resolution_amd import ddp


def test_phase_slices_tile_the_flat_gradient_buffer():
    nb = 23
    sl = ddp.phase_slices(3, 3, 64, nb, 32)
    assert len(sl) == nb + 2
    total = sum(v.numel() for v in O.make_generator_state(0).values())
    assert total == 16697987
    # phases walk the buffer from its end to its start without gaps or overlaps
    assert sl[0][1] == total and sl[-1] == (0, 64 * 3 * 9 + 64)
    for a, b in zip(sl, sl[1:]):
        assert b[1] == a[0]
    # a phase's slice is exactly its convs' (weight, bias) tensors in state_dict order
    sizes = ddp.conv_grad_sizes(3, 3, 64, nb, 32)
    names = [n for n, _ci, _co in O.generator_conv_specs()]
    assert len(sizes) == len(names) == 351
    assert sl[1][1] - sl[1][0] == sum(sizes[1 + 15 * 22: 1 + 15 * 23])


def test_bucket_plan_is_contiguous_and_ordered():
    sl = ddp.phase_slices(3, 3, 64, 23, 32)
    buckets = ddp.plan_buckets(sl, int(16 * (1 << 20) / 4))
    assert 3 <= len(buckets) <= 8
    assert buckets[0][0] == 0 and buckets[-1][1] == len(sl)
    covered = 0
    for (pb, pe, lo, hi), nxt in zip(buckets, buckets[1:] + [None]):
        assert pb < pe and lo < hi
        covered += hi - lo
        if nxt:
            assert nxt[0] == pe and nxt[3] == lo
    assert covered == 16697987
    one = ddp.plan_buckets(sl, 1 << 40)
    assert len(one) == 1 and one[0] == (0, len(sl), 0, 16697987)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nb = 3
    sl = ddp.phase_slices(3, 3, 64, nb, 32)
    buckets = ddp.plan_buckets(sl, 400_000)
    n = sl[0][1]
    flat = torch.full((n,), float("nan"))
    order = []

    def run_phases(pb, pe):          # stand-in for sr355::generator_backward_range: fills its own slices only
        for p in range(pb, pe):
            lo, hi = sl[p]
            flat[lo:hi] = (rank + 1) * (p + 1)
            order.append(p)

    ddp.reduce_in_buckets(flat, buckets, run_phases)
    expect = torch.empty(n)
    for p, (lo, hi) in enumerate(sl):
        expect[lo:hi] = (p + 1) * sum(r + 1 for r in range(world)) / world
    if rank == 0:
        ret["ok"] = bool(torch.equal(flat, expect))
        ret["order"] = order == list(range(nb + 2))
        ret["buckets"] = len(buckets)
    dist.destroy_process_group()


def test_bucketed_all_reduce_gloo():
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert ret["ok"] and ret["order"] and ret["buckets"] >= 2


def _worker_wrapped(rank, world, port, bucket_mb, ret):
    """The bucket plan of ``DistributedGenerator`` itself (3 RRDBs: 2.2 M floats) at ``bucket_mb``, ranks with
    different gradients, uneven world size."""
    import super_resolution_amd as S
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    wrapped = ddp.DistributedGenerator(S.GeneratorRRDB(num_blocks=3), bucket_mb=bucket_mb)
    sl, buckets = wrapped.slices, wrapped.buckets
    n = sl[0][1]
    gen = torch.Generator().manual_seed(100 + rank)
    mine = torch.randn(n, generator=gen)
    flat = torch.full((n,), float("nan"))
    calls = []

    def run_phases(pb, pe):
        calls.append((pb, pe))
        for p in range(pb, pe):
            lo, hi = sl[p]
            flat[lo:hi] = mine[lo:hi]

    ddp.reduce_in_buckets(flat, buckets, run_phases)
    expect = sum(torch.randn(n, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)) / world
    # the discriminator's path: start the mean, do other work, wait
    t = torch.full((1000,), float(rank + 1))
    pending = ddp.start_all_reduce_mean(t)
    other = sum(range(1000))                      # stands for the kernels enqueued while the collective runs
    pending.wait()
    if rank == world - 1:
        ret["ok"] = bool(torch.allclose(flat, expect, rtol=1e-6, atol=1e-7))
        ret["calls"] = calls == [(b[0], b[1]) for b in buckets]
        ret["buckets"] = len(buckets)
        ret["mean"] = bool(torch.equal(t, torch.full((1000,), (world + 1) / 2.0))) and other == 499500
    dist.destroy_process_group()


@pytest.mark.parametrize("bucket_mb,n_buckets", [(1.0, 4), (16.0, 1), (float("inf"), 1)])
def test_wrapped_generator_bucket_sizes_three_ranks(bucket_mb, n_buckets):
    ret = mp.Manager().dict()
    mp.spawn(_worker_wrapped, args=(3, _free_port(), bucket_mb, ret), nprocs=3, join=True)
    assert ret["ok"] and ret["calls"] and ret["mean"]
    assert ret["buckets"] == n_buckets


def _worker_direct(rank, world, port, ret):
    """The direct reduce-scatter + all-gather (two all-to-all exchanges) against the plain mean: a length that does not
    divide by the world size, the bucketed backward and the started-then-awaited form; every rank ends with the same bits."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 100_003
    mine = torch.randn(n, generator=torch.Generator().manual_seed(200 + rank))
    expect = sum(torch.randn(n, generator=torch.Generator().manual_seed(200 + r)) for r in range(world)) / world
    a = mine.clone()
    ddp.direct_mean_(a)
    b = mine.clone()
    pending = ddp.start_all_reduce_mean(b, algorithm="direct")
    pending.wait()
    # bucketed form over the generator's phase slices
    sl = ddp.phase_slices(3, 3, 64, 2, 32)
    buckets = ddp.plan_buckets(sl, 300_000)
    flat = torch.full((sl[0][1],), float("nan"))
    src = torch.randn(sl[0][1], generator=torch.Generator().manual_seed(300 + rank))

    def run_phases(pb, pe):
        for p in range(pb, pe):
            lo, hi = sl[p]
            flat[lo:hi] = src[lo:hi]

    ddp.reduce_in_buckets(flat, buckets, run_phases, algorithm="direct")
    want = sum(torch.randn(sl[0][1], generator=torch.Generator().manual_seed(300 + r)) for r in range(world)) / world
    gathered = [torch.empty_like(a) for _ in range(world)]
    dist.all_gather(gathered, a)
    if rank == 0:
        ret["mean"] = bool(torch.allclose(a, expect, rtol=1e-6, atol=1e-7)) and bool(torch.equal(a, b))
        ret["same_bits"] = all(bool(torch.equal(g, a)) for g in gathered)
        ret["buckets"] = len(buckets) >= 2 and bool(torch.allclose(flat, want, rtol=1e-6, atol=1e-7))
    dist.destroy_process_group()


def test_direct_reduce_scatter_all_gather_three_ranks():
    ret = mp.Manager().dict()
    mp.spawn(_worker_direct, args=(3, _free_port(), ret), nprocs=3, join=True)
    assert ret["mean"] and ret["same_bits"] and ret["buckets"]
