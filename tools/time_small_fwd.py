#!/usr/bin/env python3
"""GPU time of the generator forward at the training shape (16 x 3 x 32 x 32), no-grad, back to back."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import super_resolution_amd as S

dev = torch.device("cuda:0")
torch.manual_seed(0)
net = S.GeneratorRRDB().to(dev).eval()
x = torch.rand(16, 3, 32, 32, device=dev)
with torch.no_grad():
    for _ in range(5):
        net(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        net(x)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print(f"forward 16x32x32: host issue {(t1 - t0) / 50 * 1e3:.2f} ms, total {(t2 - t0) / 50 * 1e3:.2f} ms per forward")
