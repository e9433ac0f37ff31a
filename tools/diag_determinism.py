#!/usr/bin/env python3
"""Run-to-run determinism of the adversarial iteration at the training shape with the allocator's free blocks filled
with a different pattern before every run (a kernel that reads memory nothing wrote shows up as a difference).
python3 tools/diag_determinism.py [pairs]   (SR355_LIB=<other build> for comparisons)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import esrgan_oracle as O
import super_resolution_amd as S

DEV = "cuda:0"


def pollute(kind):
    blocks = []
    for mb in (512, 256, 128, 64, 32, 16, 8, 4, 2, 1):
        for _ in range(3):
            t = torch.empty(mb * 1024 * 256, dtype=torch.float32, device=DEV)
            if kind == 0:
                t.fill_(float("nan"))
            elif kind == 1:
                t.fill_(3.0e30)
            elif kind == 2:
                t.normal_()
            else:
                t.zero_()
            blocks.append(t)
    torch.cuda.synchronize()
    del blocks


def main():
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    sd_g = O.make_generator_state(101)
    sd_d = O.make_discriminator_state(102)
    sd_v = O.make_vgg_state(103)
    gen = torch.Generator().manual_seed(104)
    lr, hr = torch.rand(16, 3, 32, 32, generator=gen).to(DEV), torch.rand(16, 3, 128, 128, generator=gen).to(DEV)
    ref = None
    bad = 0
    for rep in range(pairs * 2):
        pollute(rep % 4)
        netG, netD, vgg = S.GeneratorRRDB(), S.Discriminator(), S.VGGFeatureExtractor(pretrained=False)
        netG.load_state_dict(sd_g), netD.load_state_dict(sd_d), vgg.load_state_dict(sd_v)
        netG, netD, vgg = netG.to(DEV).train(), netD.to(DEV).train(), vgg.to(DEV)
        optG = S.FlatAdam(netG.parameters(), lr=1e-4, betas=(0.9, 0.999))
        optD = S.FlatAdam(netD.parameters(), lr=1e-4, betas=(0.9, 0.999))
        out = [S.adversarial_step(netG, netD, vgg, optG, optD, lr, hr) for _ in range(2)][-1]
        torch.cuda.synchronize()
        state = {"out/" + k: v.detach().clone() for k, v in out.items()}
        state.update({"G/" + k: p.detach().clone() for k, p in netG.named_parameters()})
        state.update({"D/" + k: p.detach().clone() for k, p in netD.named_parameters()})
        if ref is None:
            ref = state
            continue
        diff = [k for k in ref if not torch.equal(ref[k], state[k])]
        if diff:
            bad += 1
            worst = max(diff, key=lambda k: (ref[k].double() - state[k].double()).abs().max().item())
            print(f"run {rep} (pattern {rep % 4}): {len(diff)} tensors differ, first {diff[:4]}, worst {worst} "
                  f"{(ref[worst].double() - state[worst].double()).abs().max().item():.3e}", flush=True)
    print(f"{os.path.basename(os.environ.get('SR355_LIB', 'default'))}: {bad} of {pairs * 2 - 1} runs differ from the first")


if __name__ == "__main__":
    main()
