import torch, time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import super_resolution_amd as S
dev="cuda:0"
gsr = S.cyclegan.GeneratorSR().to(dev).eval()
xg = torch.rand(16, 3, 64, 64, device=dev) * 2 - 1
with torch.no_grad():
    for _ in range(3): gsr(xg)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(30): gsr(xg)
    torch.cuda.synchronize(); print("GeneratorSR fwd ms", (time.perf_counter()-t0)/30*1e3)
gsr.train()
opt = torch.optim.Adam(gsr.parameters(), lr=1e-4)
def step():
    opt.zero_grad(); gsr(xg).abs().mean().backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize(); print("GeneratorSR train step ms", (time.perf_counter()-t0)/10*1e3)
