import cProfile, pstats, sys, os, torch
sys.path.insert(0, "/root/repo")
import super_resolution_amd as S
dev = torch.device("cuda:0")
netG = S.GeneratorRRDB().to(dev).train(); netD = S.Discriminator().to(dev).train(); vgg = S.VGGFeatureExtractor(pretrained=False).to(dev)
optG = S.FlatAdam(netG.parameters(), lr=1e-4); optD = S.FlatAdam(netD.parameters(), lr=1e-4)
lr = torch.rand(16, 3, 32, 32, device=dev); hr = torch.rand(16, 3, 128, 128, device=dev)
for _ in range(3): S.adversarial_step(netG, netD, vgg, optG, optD, lr, hr)
torch.cuda.synchronize()
torch.autograd.set_multithreading_enabled(False)   # backward on this thread, so that cProfile sees it
pr = cProfile.Profile(); pr.enable()
for _ in range(5): S.adversarial_step(netG, netD, vgg, optG, optD, lr, hr)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
pstats.Stats(pr).sort_stats("tottime").print_stats(45)
