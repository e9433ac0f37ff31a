import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import esrgan_oracle as O
import super_resolution_amd as S
DEV = "cuda:0"

def run(net, x, chain):
    os.environ["SR355_CHAIN"] = "1" if chain else "0"
    with torch.no_grad():
        y = net(x)
    torch.cuda.synchronize()
    return y

for blocks, kw in ((1, {}), (1, {"dense_skip_fp32": True}), (2, {}), (2, {"hr_f16": False})):
    sd = O.center_outputs(O.make_generator_state(7, num_blocks=blocks), 0.5)
    net = S.GeneratorRRDB(num_blocks=blocks, **kw); net.load_state_dict(sd); net = net.to(DEV).eval()
    x = torch.rand(8, 3, 128, 128, generator=torch.Generator().manual_seed(8)).to(DEV)
    ref = run(net, x, False)
    a = run(net, x, True)
    b = run(net, x, True)
    d = (a - ref).abs()
    print(blocks, kw, "max diff", d.max().item(), "frac bad", (d > 1e-6).float().mean().item(), "run-to-run equal", torch.equal(a, b),
          "per image", [round(v, 4) for v in d.amax(dim=(1, 2, 3)).tolist()])
    dd = torch.nan_to_num(d, nan=1e9).flatten().float()
    qs = torch.quantile(dd[::7], torch.tensor([0.5, 0.9, 0.99, 0.999], device=dd.device))
    print("   quantiles 50/90/99/99.9:", [float(v) for v in qs], "nan count", int(torch.isnan(a).sum()), "ref range", ref.min().item(), ref.max().item())
    if d.max().item() > 0:
        bad = (d[0, 0] > 1e-6)
        ys, xs = bad.nonzero(as_tuple=True)
        print("   image 0 bad rows", ys.min().item(), ys.max().item(), "cols", xs.min().item(), xs.max().item(), "count", bad.sum().item())
