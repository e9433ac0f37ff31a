import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.nn.functional as F
from oracle import esrgan_oracle as O
import super_resolution_amd as S
DEV = "cuda:0"
g = dict(np.load("tests/golden/g3_gen2.npz"))
t = lambda a: torch.from_numpy(np.asarray(a))
x = t(g["x"])
print("SR355_DENSE_FUSED", os.environ.get("SR355_DENSE_FUSED"))
base = O.center_outputs(O.make_generator_state(int(g["seed"]), num_blocks=1), 0.5)
def variant(level):
    sd = {k: v.clone() for k, v in base.items()}
    if level >= 1:
        for k in sd:
            if ".conv5." in k and not k.startswith("body.0.rdb1"): sd[k].zero_()
    if level >= 2:
        for k in sd:
            if ".conv5." in k: sd[k].zero_()
    return sd
for level in (0, 1, 2):
    sd = variant(level)
    for kw, mode in ((dict(hr_split=False), True),):
        for grad in (True, False):
            net = S.GeneratorRRDB(num_blocks=1, hr_f16=False, dense_skip_fp32=True, **kw); net.load_state_dict(sd); net = net.to(DEV).train()
            with torch.set_grad_enabled(grad):
                out = net(x.to(DEV)).detach().cpu()
            e, f = O.generator_forward(sd, x, emulate_bf16=mode), O.generator_forward(sd, x)
            print("level", level, kw, grad, "vs emulated %.2e  vs fp32 %.2e  emul-vs-fp32 %.2e" % ((out - e).abs().mean().item(), (out - f).abs().mean().item(), (e - f).abs().mean().item()))
