#!/usr/bin/env python3
"""Time of the training-shape forward pass (16 x 3 x 32 x 32, recorded for a backward) by HIP events, steps back to
back (SR355_LIB=<other build> for same-box comparisons).  python3 tools/time_train_forward.py [steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import super_resolution_amd as S


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = S.GeneratorRRDB().to(dev).train()
    x = torch.rand(16, 3, 32, 32, device=dev)
    for _ in range(5):
        net(x)
    best = 1e9
    for _rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(steps):
            net(x)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / steps)
    print(f"{os.path.basename(os.environ.get('SR355_LIB', 'default')):>24s}: training forward {best:.3f} ms")


if __name__ == "__main__":
    main()
