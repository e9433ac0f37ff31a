"""Data-parallel training of the generator: one process per GPU, bucketed gradient all-reduce over
RCCL overlapped with the backward pass.

The reference trains under ``nn.DataParallel`` (/root/reference/esrgan/train.py:108-111): one process,
parameters broadcast every forward, gradients reduced onto GPU 0 after the whole backward.  Here every
rank keeps its own replica and its own batch; the backward pass runs in ordered phases
(``sr355_generator_backward_range``) and as soon as a phase group has written its slice of the flat
gradient buffer that slice is all-reduced asynchronously while the next group computes.  Gradients of
the HR-side convs come first, then RRDB 22 .. 0, then conv_first: the buckets are contiguous slices
taken from the END of the flat buffer towards its start.

The mean over ranks equals the reference's mean over the gathered batch when ranks hold equal batches.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn
from torch import Tensor


def conv_grad_sizes(in_channels: int, out_channels: int, nf: int, nb: int, gc: int) -> List[int]:
    """Floats of (weight + bias) gradient per conv in state_dict order."""
    sizes = [nf * in_channels * 9 + nf]
    for _ in range(3 * nb):
        for k in range(5):
            cin, cout = nf + k * gc, (gc if k < 4 else nf)
            sizes.append(cout * cin * 9 + cout)
    sizes += [nf * nf * 9 + nf] * 4
    sizes.append(out_channels * nf * 9 + out_channels)
    return sizes


def phase_slices(in_channels: int, out_channels: int, nf: int, nb: int, gc: int) -> List[Tuple[int, int]]:
    """[lo, hi) float range of the flat gradient buffer written by each backward phase."""
    sizes = conv_grad_sizes(in_channels, out_channels, nf, nb, gc)
    starts = [0]
    for s in sizes:
        starts.append(starts[-1] + s)
    out = [(starts[1 + 15 * nb], starts[-1])]                         # phase 0: trunk_conv .. conv_last
    for p in range(1, nb + 1):                                        # phase p: RRDB nb - p
        b = nb - p
        out.append((starts[1 + 15 * b], starts[1 + 15 * (b + 1)]))
    out.append((0, starts[1]))                                        # last phase: conv_first
    return out


def plan_buckets(slices: Sequence[Tuple[int, int]], bucket_floats: int) -> List[Tuple[int, int, int, int]]:
    """Group consecutive phases into buckets of at least ``bucket_floats`` floats:
    (phase_begin, phase_end, lo, hi) with [lo, hi) contiguous in the flat buffer."""
    buckets, begin = [], 0
    while begin < len(slices):
        end, lo, hi = begin, slices[begin][0], slices[begin][1]
        while True:
            end += 1
            if end >= len(slices) or hi - lo >= bucket_floats:
                break
            lo = min(lo, slices[end][0])
            hi = max(hi, slices[end][1])
        buckets.append((begin, end, lo, hi))
        begin = end
    # phases walk the buffer backwards, so every bucket must be one contiguous range
    for (_b, _e, lo, hi), (pb, pe) in zip(buckets, [(b[0], b[1]) for b in buckets]):
        assert sum(s[1] - s[0] for s in slices[pb:pe]) == hi - lo
    return buckets


def reduce_in_buckets(flat: Tensor, buckets: Sequence[Tuple[int, int, int, int]],
                      run_phases: Callable[[int, int], None], group=None, algorithm: str = "all_reduce") -> None:
    """Run the phase groups in order; after each, start the all-reduce (mean) of its slice and go on."""
    import torch.distributed as dist

    if algorithm == "direct":
        pending = []
        for pb, pe, lo, hi in buckets:
            run_phases(pb, pe)
            pending.append(PendingMean(flat[lo:hi], group, "direct"))
        for p in pending:
            p.wait()
        return
    world = dist.get_world_size(group)
    avg = dist.get_backend(group) == "nccl"
    handles = []
    for pb, pe, lo, hi in buckets:
        run_phases(pb, pe)
        view = flat[lo:hi]
        if avg:
            handles.append((dist.all_reduce(view, op=dist.ReduceOp.AVG, group=group, async_op=True), view, None))
        elif view.is_cuda:
            # gloo (tests, CPU-only fabrics): device tensors are staged through host memory
            host = view.cpu()
            handles.append((dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group, async_op=True), view, host))
        else:
            handles.append((dist.all_reduce(view, op=dist.ReduceOp.SUM, group=group, async_op=True), view, None))
    for h, view, host in handles:
        h.wait()
        if host is not None:
            view.copy_(host)
        if not avg:
            view.div_(world)


_COMM_STREAMS = {}


def _comm_stream(device: torch.device) -> "torch.cuda.Stream":
    """One side stream per device for collectives that are chains of operations (kept for the life of the process)."""
    s = _COMM_STREAMS.get(device)
    if s is None:
        s = _COMM_STREAMS[device] = torch.cuda.Stream(device=device)
    return s


def direct_mean_(buf: Tensor, group=None) -> None:
    """In-place mean of the flat tensor ``buf`` over the ranks of ``group`` as a direct reduce-scatter + all-gather:
    two all-to-all exchanges.  In the first every rank sends chunk j of its buffer to rank j (all peers at once: on a
    fully connected xGMI node every link carries 1/world of the buffer, where a ring pushes 2 (world-1)/world of it
    through each link in turn); rank j adds the world chunks it now holds in rank order and divides; in the second
    every rank sends its reduced chunk to all peers.  Each element is reduced on exactly one rank in a fixed order, so
    all ranks end with the same bits.  Blocking with respect to the current stream / the calling thread."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    n = buf.numel()
    if world == 1 or n == 0:
        return
    flat = buf.view(-1)
    chunk = -(-n // world)
    if chunk * world == n:
        send = flat
    else:
        send = flat.new_zeros(chunk * world)
        send[:n].copy_(flat)
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send, group=group)
    parts = recv.view(world, chunk)
    red = parts[0].clone()
    for r in range(1, world):
        red.add_(parts[r])
    red.div_(world)
    dist.all_to_all_single(recv, red.repeat(world), group=group)
    flat.copy_(recv[:n])


class PendingMean:
    """An in-place mean over the ranks of a group that has been started and not yet waited for.  ``algorithm``:
    ``"all_reduce"`` (one RCCL all-reduce, the default) or ``"direct"`` (``direct_mean_`` on a side stream)."""

    def __init__(self, t: Tensor, group=None, algorithm: str = "all_reduce"):
        import torch.distributed as dist
        if algorithm not in ("all_reduce", "direct"):
            raise ValueError("algorithm must be 'all_reduce' or 'direct'")
        self.t, self.world = t, dist.get_world_size(group)
        self.host = None
        self.event = None
        if algorithm == "direct":
            self.handle, self.avg = None, True
            if dist.get_backend(group) == "nccl":
                # the chain (exchange, add, exchange, copy) runs on a side stream behind what the caller's stream has
                # enqueued so far; wait() makes the caller's stream wait for its end
                side = _comm_stream(t.device)
                side.wait_stream(torch.cuda.current_stream(t.device))
                t.record_stream(side)
                with torch.cuda.stream(side):
                    direct_mean_(t, group)
                    self.event = torch.cuda.Event()
                    self.event.record(side)
            else:
                self.host = t.cpu() if t.is_cuda else t
                direct_mean_(self.host, group)
            return
        if dist.get_backend(group) == "nccl":
            # RCCL runs the collective on its own stream; wait() makes the caller's stream (not the host) wait for it
            self.handle = dist.all_reduce(t, op=dist.ReduceOp.AVG, group=group, async_op=True)
            self.avg = True
        else:
            # gloo (tests, CPU-only fabrics): a device tensor is staged through host memory (gloo's own device-tensor
            # collectives stall when ranks share a GPU)
            self.host = t.cpu() if t.is_cuda else t
            self.handle = dist.all_reduce(self.host, op=dist.ReduceOp.SUM, group=group, async_op=True)
            self.avg = False

    def wait(self) -> None:
        if self.handle is None:          # the direct form
            if self.event is not None:
                torch.cuda.current_stream(self.t.device).wait_event(self.event)
            elif self.host is not self.t:
                self.t.copy_(self.host)
            return
        self.handle.wait()
        if not self.avg:
            self.host.div_(self.world)
            if self.host is not self.t:
                self.t.copy_(self.host)


def start_all_reduce_mean(t: Tensor, group=None, algorithm: str = "all_reduce") -> PendingMean:
    """Start the in-place mean of ``t`` over the ranks of ``group`` and return at once; kernels enqueued before
    ``wait()`` on the returned object overlap with the collective."""
    return PendingMean(t, group, algorithm)


def all_reduce_mean(t: Tensor, group=None, algorithm: str = "all_reduce") -> None:
    """In-place mean of ``t`` over the ranks of ``group``: one RCCL all-reduce (``backend="nccl"``), or the direct
    reduce-scatter + all-gather of ``direct_mean_``."""
    PendingMean(t, group, algorithm).wait()


def broadcast_tensors(tensors, src: int = 0, group=None) -> None:
    """Rank ``src``'s values into every rank's tensors (parameters and buffers of a replica at start-up)."""
    import torch.distributed as dist
    staged = dist.get_backend(group) != "nccl"
    for t in tensors:
        if staged and t.is_cuda:
            host = t.detach().cpu()
            dist.broadcast(host, src=src, group=group)
            t.detach().copy_(host)
        else:
            dist.broadcast(t.detach(), src=src, group=group)


class DistributedGenerator(nn.Module):
    """Wraps a ``GeneratorRRDB`` replica.  ``forward`` is the module's own; in grad mode the backward pass
    all-reduces the gradients bucket by bucket while it runs.  Call ``broadcast_parameters()`` once after
    construction so that every rank starts from rank 0's weights."""

    def __init__(self, module: nn.Module, bucket_mb: float = 16.0, group=None, algorithm: str = "all_reduce"):
        super().__init__()
        self.module = module
        self.group = group
        self.algorithm = algorithm      # "all_reduce" | "direct" (see direct_mean_)
        m = module
        self.slices = phase_slices(m.in_channels, m.out_channels, m.num_features, m.num_blocks, m.growth_channels)
        # bucket_mb = float("inf"): one all-reduce of the whole buffer after the backward
        self.buckets = plan_buckets(self.slices, (1 << 62) if bucket_mb == float("inf") else int(bucket_mb * (1 << 20) / 4))

    def broadcast_parameters(self, src: int = 0) -> None:
        import torch.distributed as dist
        staged = dist.get_backend(self.group) != "nccl"
        for p in self.module.parameters():
            if staged and p.is_cuda:
                host = p.data.cpu()
                dist.broadcast(host, src=src, group=self.group)
                p.data.copy_(host)
            else:
                dist.broadcast(p.data, src=src, group=self.group)
        # writes through .data do not move the parameters' version counters, which is what the weight-pack caches look at
        torch.autograd.graph.increment_version(list(self.module.parameters()))

    def forward(self, x: Tensor) -> Tensor:
        from .esrgan_model import _GeneratorFunction, _graph_params
        m = self.module
        xin = x.float().contiguous()
        params = [p for c in m.conv_modules() for p in (c.weight, c.bias)]
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            return _GeneratorFunction.apply(xin, m, self, *_graph_params(m, params))
        return m(x)
