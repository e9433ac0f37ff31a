#!/usr/bin/env python3
"""Run-to-run reproducibility of the pieces of an adversarial iteration at the training shape (VGG / D / G forward and
backward, 30 repetitions each, the allocator's free blocks refilled with noise every five): which piece differs
from its first run.  python3 tools/diag_pieces.py   (SR355_LIB=<other build> for comparisons)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import esrgan_oracle as O
import super_resolution_amd as S
DEV="cuda:0"
torch.manual_seed(0)
sd_g = O.make_generator_state(101); sd_d = O.make_discriminator_state(102); sd_v = O.make_vgg_state(103)
netG, netD, vgg = S.GeneratorRRDB(), S.Discriminator(), S.VGGFeatureExtractor(pretrained=False)
netG.load_state_dict(sd_g); netD.load_state_dict(sd_d); vgg.load_state_dict(sd_v)
netG, netD, vgg = netG.to(DEV).train(), netD.to(DEV).train(), vgg.to(DEV)
gen = torch.Generator().manual_seed(104)
lr, hr = torch.rand(16,3,32,32,generator=gen).to(DEV), torch.rand(16,3,128,128,generator=gen).to(DEV)
def junk():
    t=[torch.empty(s*1024*256,device=DEV).normal_() for s in (256,64,16,4,1)]; torch.cuda.synchronize(); del t
def rep(name, fn, n=30):
    ref=None; bad=0
    for i in range(n):
        if i%5==0: junk()
        out=fn(); torch.cuda.synchronize()
        out=[o.detach().clone() for o in out]
        if ref is None: ref=out
        elif any(not torch.equal(a,b) for a,b in zip(ref,out)): bad+=1
    print(f"{name}: {bad} of {n-1} differ", flush=True)
def vgg_fwd():
    with torch.no_grad(): return [vgg(hr)]
def d_fwd():
    with torch.no_grad(): return [netD(hr)]
def g_fwd():
    with torch.no_grad(): return [netG(lr)]
def g_train_fwd():
    return [netG(lr)]
def vgg_bwd():
    x=hr.clone().requires_grad_(True); vgg(x).square().mean().backward(); return [x.grad]
def d_bwd():
    netD.zero_grad(set_to_none=True); x=hr.clone().requires_grad_(True); netD(x).square().mean().backward()
    return [x.grad]+[p.grad for p in netD.parameters()]
def g_bwd():
    netG.zero_grad(set_to_none=True); (netG(lr)-hr).abs().mean().backward(); return [p.grad for p in netG.parameters()]
for name,fn in [("vgg forward",vgg_fwd),("D forward (train BN)",d_fwd),("G no-grad forward",g_fwd),("G recorded forward",g_train_fwd),("vgg backward",vgg_bwd),("D backward",d_bwd),("G backward",g_bwd)]:
    rep(name,fn)
