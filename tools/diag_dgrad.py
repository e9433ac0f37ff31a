#!/usr/bin/env python3
"""Where the discriminator's like-for-like gradient error comes from: the MSE-GAN cotangent (a function of the
prediction, nearly constant over the map) against a fixed white-noise cotangent, HIP path vs the oracle with bf16 conv
operands; and how far the two forward passes are apart relative to the prediction's own spread."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import esrgan_oracle as O
import super_resolution_amd as S
DEV = "cuda:0"
def rel(a, b): return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
sd = O.make_discriminator_state(81)
gen = torch.Generator().manual_seed(82)
for k in list(sd):
    if k.endswith("running_mean"):
        ci = k.rsplit(".", 1)[0]
        sd[ci + ".weight"] = 1.0 + 0.5 * torch.randn(sd[k].shape, generator=gen)
        sd[ci + ".bias"] = 0.3 * torch.randn(sd[k].shape, generator=gen)
img = torch.rand(4, 3, 64, 64, generator=gen)
r = torch.randn(4, 1, 4, 4, generator=gen)
pre_bn = {f"model.{ci}.bias" for ci, _a, _b, _s, bi in O.discriminator_layout() if bi is not None}
for name, lossf in (("mse-gan cotangent", lambda p, rr: torch.nn.functional.mse_loss(p, torch.ones_like(p))),
                    ("white-noise cotangent", lambda p, rr: (p * rr).sum())):
    netD = S.Discriminator(); netD.load_state_dict(sd); netD = netD.to(DEV).train()
    xg = img.to(DEV).requires_grad_(True)
    pred = netD(xg); lossf(pred, r.to(DEV)).backward()
    leaves = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd.items()}
    xr = img.clone().requires_grad_(True)
    pred_r = O.discriminator_forward(leaves, xr, True, emulate_bf16=True); lossf(pred_r, r).backward()
    errs = sorted(((rel(p.grad.cpu(), leaves[k].grad), k) for k, p in netD.named_parameters() if k not in pre_bn), reverse=True)
    print(f"{name}: worst {errs[0]}, median {errs[len(errs)//2][0]:.4f}, input {rel(xg.grad.cpu(), xr.grad):.4f}; "
          f"forward: |pred - pred_oracle| rms {(pred.detach().cpu()-pred_r.detach()).pow(2).mean().sqrt().item():.3e}, "
          f"pred mean {pred_r.mean().item():.3f} std {pred_r.std().item():.3e}")
