#!/usr/bin/env python3
"""The adversarial training iteration a few times (for rocprofv3 passes and wall-clock timing)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import super_resolution_amd as S


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    netG = S.GeneratorRRDB().to(dev).train()
    netD = S.Discriminator().to(dev).train()
    vgg = S.VGGFeatureExtractor(pretrained=False).to(dev)
    optG = S.FlatAdam(netG.parameters(), lr=1e-4)
    optD = S.FlatAdam(netD.parameters(), lr=1e-4)
    lr = torch.rand(16, 3, 32, 32, device=dev)
    hr = torch.rand(16, 3, 128, 128, device=dev)
    for _ in range(2):
        S.adversarial_step(netG, netD, vgg, optG, optD, lr, hr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        S.adversarial_step(netG, netD, vgg, optG, optD, lr, hr)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"adversarial step: host issue {(t1 - t0) / steps * 1e3:.2f} ms, total {(t2 - t0) / steps * 1e3:.2f} ms")


if __name__ == "__main__":
    main()
