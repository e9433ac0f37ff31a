#!/usr/bin/env python3
"""Which launch of a discriminator forward + backward first differs from the first repetition.

Every op of conv_core (layout kernels, conv forward, BatchNorm forward / backward, conv data / weight gradients) is
wrapped so that its outputs are copied aside; after each repetition the copies are compared with the first
repetition's, in launch order, and the first differing tensor is described (how many elements, where in the blocked
layout, how far).  The allocator's free blocks are refilled with noise every third repetition.

python3 tools/diag_layers.py [reps] [guard]   (SR355_LIB=<other build> for comparisons; guard=1: NaN guard bands)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import esrgan_oracle as O
import super_resolution_amd as S
from super_resolution_amd import conv_core as cc
from super_resolution_amd import ops

DEV = "cuda:0"
LOG = []


def _keep(name, *tensors):
    for k, t in enumerate(tensors):
        if t is not None:
            LOG.append((f"{name}/{k}", t.detach().clone()))


def _wrap():
    cf, cb = cc._conv_forward, cc._conv_backward
    bf, bb = ops.bn_lrelu_fwd_now, ops.bn_lrelu_bwd_now
    tb, fb = ops.to_blocked_now, ops.from_blocked_now

    def conv_forward(x, weight, bias, slope, out_kind, stride):
        y = cf(x, weight, bias, slope, out_kind, stride)
        _keep(f"conv_fwd {tuple(weight.shape[:2])} s{stride} {out_kind}", y)
        return y

    def conv_backward(x, weight, bias, dyb, stride, need_dx, need_dw):
        r = cb(x, weight, bias, dyb, stride, need_dx, need_dw)
        _keep(f"conv_bwd {tuple(weight.shape[:2])} s{stride}", *r)
        return r

    def bn_fwd(*a, **k):
        r = bf(*a, **k)
        _keep(f"bn_fwd C={a[10].numel()}", a[10], a[11], a[12])
        return r

    def bn_bwd(*a, **k):
        r = bb(*a, **k)
        _keep(f"bn_bwd C={a[9].numel()}", a[8], a[9], a[10])
        return r

    def to_blocked(*a, **k):
        r = tb(*a, **k)
        _keep("to_blocked", r)
        return r

    def from_blocked(*a, **k):
        r = fb(*a, **k)
        _keep("from_blocked", r)
        return r

    cc._conv_forward, cc._conv_backward = conv_forward, conv_backward
    ops.bn_lrelu_fwd_now, ops.bn_lrelu_bwd_now = bn_fwd, bn_bwd
    ops.to_blocked_now, ops.from_blocked_now = to_blocked, from_blocked


def junk():
    t = [torch.empty(mb * 1024 * 256, device=DEV).normal_() for mb in (256, 64, 16, 4, 1)]
    torch.cuda.synchronize()
    del t


def describe(name, a, b):
    ne = (a != b) if a.dtype != torch.bfloat16 else (a.view(torch.int16) != b.view(torch.int16))
    idx = ne.nonzero()
    d = (a.double() - b.double()).abs()
    print(f"   {name}: shape {tuple(a.shape)} {a.dtype}, {idx.shape[0]} of {a.numel()} elements differ, "
          f"max |diff| {d.max().item():.3e} (max |ref| {a.double().abs().max().item():.3e})", flush=True)
    for dim in range(a.dim()):
        vals = idx[:, dim].unique().tolist()
        print(f"      dim {dim}: {len(vals)} distinct, {vals[:40]}{' ...' if len(vals) > 40 else ''}", flush=True)
    for row in idx[:12].tolist():
        print(f"      {row}: ref {a[tuple(row)].item()!r} got {b[tuple(row)].item()!r}", flush=True)


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    _wrap()
    netD = S.Discriminator()
    netD.load_state_dict(O.make_discriminator_state(102))
    netD = netD.to(DEV).train()
    hr = torch.rand(16, 3, 128, 128, generator=torch.Generator().manual_seed(104)).to(DEV)
    ref, bad = None, 0
    for rep in range(reps):
        if rep % 3 == 0:
            junk()
        LOG.clear()
        netD.zero_grad(set_to_none=True)
        x = hr.clone().requires_grad_(True)
        netD(x).square().mean().backward()
        torch.cuda.synchronize()
        got = list(LOG)
        if ref is None:
            ref = got
            print(f"{len(ref)} recorded tensors per repetition: " + "; ".join(n for n, _ in ref), flush=True)
            continue
        first = next((i for i, ((_, a), (_, b)) in enumerate(zip(ref, got)) if not torch.equal(a, b)), None)
        if first is not None:
            bad += 1
            n_diff = sum(1 for (_, a), (_, b) in zip(ref, got) if not torch.equal(a, b))
            print(f"rep {rep} (junk {'just before' if rep % 3 == 0 else 'earlier'}): {n_diff} tensors differ, first is "
                  f"#{first}", flush=True)
            describe(ref[first][0], ref[first][1], got[first][1])
            if bad >= 6:
                break
    print(f"{os.path.basename(os.environ.get('SR355_LIB', 'default'))}: {bad} of {rep} repetitions differ from the first",
          flush=True)


if __name__ == "__main__":
    main()
