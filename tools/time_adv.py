#!/usr/bin/env python3
"""Wall-clock time of the adversarial training iteration at the benchmark's training shape, steps issued back to back.
python3 tools/time_adv.py [steps] [pair|separate|dsep]   (separate: the two VGG19 forwards and the two discriminator passes
of the D update as separate calls; dsep: only the discriminator passes separate)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import super_resolution_amd as S
from super_resolution_amd import esrgan_aux


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    mode = sys.argv[2] if len(sys.argv) > 2 else "pair"
    if mode == "separate":
        esrgan_aux._perceptual_features = lambda vgg, fake, hr: (vgg(fake), vgg(hr))
    if mode in ("separate", "dsep"):
        S.Discriminator.forward_pair = None
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    netG = S.GeneratorRRDB().to(dev).train()
    netD = S.Discriminator().to(dev).train()
    vgg = S.VGGFeatureExtractor(pretrained=False).to(dev)
    og = S.FlatAdam(netG.parameters(), lr=1e-4)
    od = S.FlatAdam(netD.parameters(), lr=1e-4)
    lr = torch.rand(16, 3, 32, 32, device=dev)
    hr = torch.rand(16, 3, 128, 128, device=dev)
    for _ in range(3):
        S.adversarial_step(netG, netD, vgg, og, od, lr, hr)
    best = 1e9
    for _rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            S.adversarial_step(netG, netD, vgg, og, od, lr, hr)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps * 1e3)
    print(f"{mode:>10s}: adversarial step {best:.3f} ms")


if __name__ == "__main__":
    main()
