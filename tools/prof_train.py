#!/usr/bin/env python3
"""The generator's pixel-loss training step a few times (for rocprofv3 passes and wall-clock timing).

    python3 tools/prof_train.py [steps] [batch] [lr_size]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import super_resolution_amd as S


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    L = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = S.GeneratorRRDB().to(dev).train()
    opt = S.FlatAdam(net.parameters(), lr=1e-4)          # torch.optim.Adam's update in one launch
    lr = torch.rand(B, 3, L, L, device=dev)
    hr = torch.rand(B, 3, 4 * L, 4 * L, device=dev)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.l1_loss(net(lr), hr)
        loss.backward()
        opt.step()
        return loss

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    marks = []
    for _ in range(steps):
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        out = net(lr)
        t1 = time.perf_counter()
        loss = torch.nn.functional.l1_loss(out, hr)
        loss.backward()
        t2 = time.perf_counter()
        opt.step()
        t3 = time.perf_counter()
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        marks.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0))
    m = [sum(x[i] for x in marks) / len(marks) * 1e3 for i in range(5)]
    print(f"host ms: forward issue {m[0]:.2f}, backward issue {m[1]:.2f}, adam issue {m[2]:.2f}, drain {m[3]:.2f}, "
          f"step {m[4]:.2f}  ->  {B / m[4] * 1e3:.1f} patches/s")


if __name__ == "__main__":
    main()
