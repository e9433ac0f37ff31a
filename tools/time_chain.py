#!/usr/bin/env python3
"""Forward time of the benchmark batch and of its chain launch, for same-box comparisons of library builds
(SR355_LIB=<other build> python3 tools/time_chain.py [batch] [size] [steps])."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import super_resolution_amd as S
from super_resolution_amd import _lib


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = S.GeneratorRRDB().to(dev).eval()
    x = torch.rand(B, 3, L, L, device=dev)
    zeros = os.environ.get("TIME_CHAIN_ZEROS", "0")
    if zeros in ("1", "2"):      # 1: all-zero operands (the same traffic at the clock an idle matrix pipe allows); 2: zero
        with torch.no_grad():    # weights only: random activations move through LDS and the fabric, the products are zero
            for p in net.parameters():
                p.zero_()
        if zeros == "1":
            x.zero_()
    if zeros == "3":             # 3: zero body weights, ones as biases: constant non-zero activations
        with torch.no_grad():
            for n_, p in net.named_parameters():
                p.zero_() if n_.endswith("weight") else p.fill_(1.0)
    L_ = _lib.lib()
    L_.sr355_chain_timing(1)
    L_.sr355_chain_last_ms.restype = __import__("ctypes").c_float
    with torch.no_grad():
        for _ in range(5):
            net(x)
        torch.cuda.synchronize()
        chain = []
        t0 = time.perf_counter()
        for _ in range(steps):
            net(x)
            torch.cuda.synchronize()
            chain.append(L_.sr355_chain_last_ms())
        t1 = time.perf_counter()
    chain.sort()
    print(f"{os.environ.get('SR355_LIB', 'default'):>60s}: forward {(t1 - t0) / steps * 1e3:.3f} ms (synchronised per step), "
          f"chain median {chain[len(chain) // 2]:.3f} ms, min {chain[0]:.3f}")


if __name__ == "__main__":
    main()
