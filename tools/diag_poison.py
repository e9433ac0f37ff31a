#!/usr/bin/env python3
"""Every buffer torch hands out uninitialised filled with NaN bit patterns (torch.use_deterministic_algorithms +
fill_uninitialized_memory): a kernel that reads memory nothing wrote turns its outputs into NaN or changes them.
Runs the inference forward, the pixel iteration and the adversarial iteration with and without the poison and
compares.  python3 tools/diag_poison.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import esrgan_oracle as O
import super_resolution_amd as S

DEV = "cuda:0"


def run_all():
    out = {}
    sd_g = O.make_generator_state(101)
    sd_d = O.make_discriminator_state(102)
    sd_v = O.make_vgg_state(103)
    gen = torch.Generator().manual_seed(104)
    lr, hr = torch.rand(16, 3, 32, 32, generator=gen).to(DEV), torch.rand(16, 3, 128, 128, generator=gen).to(DEV)
    big = torch.rand(16, 3, 128, 128, generator=gen).to(DEV)
    netG, netD, vgg = S.GeneratorRRDB(), S.Discriminator(), S.VGGFeatureExtractor(pretrained=False)
    netG.load_state_dict(sd_g), netD.load_state_dict(sd_d), vgg.load_state_dict(sd_v)
    netG, netD, vgg = netG.to(DEV), netD.to(DEV).train(), vgg.to(DEV)
    with torch.no_grad():
        out["infer 16x128^2"] = netG.eval()(big).clone()
        out["infer 2x3x37x53"] = netG(big[:2, :, :37, :53].contiguous()).clone()
    netG.train()
    optG = S.FlatAdam(netG.parameters(), lr=1e-4)
    optD = S.FlatAdam(netD.parameters(), lr=1e-4)
    for _ in range(2):
        r = S.pixel_step(netG, optG, lr, hr)
    out.update({"pixel/" + k: v.detach().clone() for k, v in (r.items() if isinstance(r, dict) else {"loss": r}.items())})
    for _ in range(2):
        r = S.adversarial_step(netG, netD, vgg, optG, optD, lr, hr)
    out.update({"adv/" + k: v.detach().clone() for k, v in r.items()})
    out.update({"G/" + k: p.detach().clone() for k, p in netG.named_parameters()})
    out.update({"D/" + k: p.detach().clone() for k, p in netD.named_parameters()})
    torch.cuda.synchronize()
    return out


def main():
    ref = run_all()
    torch.use_deterministic_algorithms(True, warn_only=True)
    torch.utils.deterministic.fill_uninitialized_memory = True
    got = run_all()
    bad_nan = [k for k, v in got.items() if not torch.isfinite(v.float()).all()]
    diff = [k for k in ref if not torch.equal(ref[k], got[k])]
    print("non-finite under poison:", len(bad_nan), bad_nan[:8])
    print("different under poison:", len(diff), diff[:8])
    for k in diff[:4]:
        d = (ref[k].float() - got[k].float()).abs()
        print("   ", k, "max abs diff", d.max().item(), "of max", ref[k].float().abs().max().item())


if __name__ == "__main__":
    main()
