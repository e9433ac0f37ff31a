#!/usr/bin/env python3
"""Time of one 64 -> 64 conv at the up-sampling side's full size (16 x 512 x 512, bf16 blocked in and out) by HIP
events (SR355_LIB=<other build> for same-box comparisons).  python3 tools/time_hr_conv.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from super_resolution_amd import ops
from super_resolution_amd._lib import DTYPE_BF16


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    N, H, W = 16, 512, 512
    xb = torch.randn(N, H, 8, W, 8, generator=g).to(dev).to(torch.bfloat16)
    w = (torch.randn(64, 64, 3, 3, generator=g) * 0.05).to(dev)
    b = torch.randn(64, generator=g).to(dev)
    wp, bp = ops.pack_weight(w, b, DTYPE_BF16)
    act = torch.zeros(N, H, 8, W, 8, dtype=torch.bfloat16, device=dev)

    def run():
        ops.conv3x3_fwd(xb, 8, wp, bp, 64, 0.2, 0, None, 0.0, None, 0.0, act, 0, None, None)

    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    gb = 2 * N * H * W * 64 * 2 / 1e9
    print(f"{os.path.basename(os.environ.get('SR355_LIB', 'default')):>22s}: {us:7.1f} us  {gb / us * 1e6 / 1e3:5.2f} TB/s  {N * H * W * 64 * 576 * 2 / us / 1e6:6.1f} TFLOP/s")


if __name__ == "__main__":
    main()
