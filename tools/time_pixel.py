#!/usr/bin/env python3
"""Wall-clock time of the pixel-loss and adversarial training iterations at the benchmark's training shape, steps
issued back to back (SR355_LIB=<other build> for same-box comparisons).  python3 tools/time_pixel.py [steps] [adv]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import super_resolution_amd as S


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = S.GeneratorRRDB().to(dev).train()
    opt = S.FlatAdam(net.parameters(), lr=1e-4)
    lr = torch.rand(16, 3, 32, 32, device=dev)
    hr = torch.rand(16, 3, 128, 128, device=dev)
    for _ in range(3):
        S.pixel_step(net, opt, lr, hr)
    best = 1e9
    for _rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            S.pixel_step(net, opt, lr, hr)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps * 1e3)
    print(f"{os.environ.get('SR355_LIB', 'default'):>50s}: pixel step {best:.3f} ms")


if __name__ == "__main__":
    main()
