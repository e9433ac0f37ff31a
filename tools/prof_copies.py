#!/usr/bin/env python3
"""Where the small device copies of a pixel training iteration come from: aten::copy_ / fill_ / zero_ calls of one
iteration grouped by the Python line that issued them.  python3 tools/prof_copies.py"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import super_resolution_amd as S


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = S.GeneratorRRDB().to(dev).train()
    opt = S.FlatAdam(net.parameters(), lr=1e-4)
    lr = torch.rand(16, 3, 32, 32, device=dev)
    hr = torch.rand(16, 3, 128, 128, device=dev)
    for _ in range(3):
        S.pixel_step(net, opt, lr, hr)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        S.pixel_step(net, opt, lr, hr)
        torch.cuda.synchronize()
    sites = collections.Counter()
    for ev in prof.events():
        if ev.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::clone", "aten::contiguous", "aten::to"):
            stack = [s for s in (ev.stack or []) if "super_resolution_amd" in s or "tools/" in s]
            sites[(ev.name, stack[0] if stack else "?")] += 1
    for (name, site), n in sites.most_common(25):
        print(f"{n:5d}  {name:18s} {site}")
    kernels = collections.Counter(ev.name for ev in prof.events() if ev.device_type == torch.autograd.DeviceType.CUDA)
    for k, n in kernels.most_common(12):
        print(f"{n:5d}  {k[:100]}")


if __name__ == "__main__":
    main()
