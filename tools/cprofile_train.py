#!/usr/bin/env python3
"""Host profile of the pixel training iteration with the backward on the calling thread (so cProfile sees it)."""
import cProfile, os, pstats, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import super_resolution_amd as S

dev = torch.device("cuda:0")
net = S.GeneratorRRDB().to(dev).train()
opt = S.FlatAdam(net.parameters(), lr=1e-4)
lr = torch.rand(16, 3, 32, 32, device=dev)
hr = torch.rand(16, 3, 128, 128, device=dev)
torch.autograd.set_multithreading_enabled(False)


def step():
    opt.zero_grad(set_to_none=True)
    torch.nn.functional.l1_loss(net(lr), hr).backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
