#!/usr/bin/env python3
"""Layer by layer: the discriminator's backward on the HIP path against autograd through the oracle's forward with
bf16 conv operands, under a fixed white-noise cotangent: relative L2 error of the gradient w.r.t. every conv output
(what BatchNorm's backward hands to the conv) and every conv input (what the conv's data gradient hands down)."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import esrgan_oracle as O
import super_resolution_amd as S
from super_resolution_amd import conv_core as cc, ops
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
PIN = len(sys.argv) > 1 and sys.argv[1] == "pin"
# --- HIP path first when the LeakyReLU slopes are pinned to its forward pass (argument "pin")
masks = {}
if PIN:
    netP = S.Discriminator(); netP.load_state_dict(sd); netP = netP.to(DEV).train()
    cf = cc._conv_forward
    seen = []
    def conv_forward(x_, *a, **k):
        seen.append(x_.detach().clone())
        return cf(x_, *a, **k)
    cc._conv_forward = conv_forward
    netP(img.to(DEV).requires_grad_(True))
    cc._conv_forward = cf
    for j in range(1, len(seen)):          # input of conv j = activation behind conv j - 1
        masks[j - 1] = (cc.blocked_as_nchw(seen[j]).float().cpu() > 0)
# --- oracle with retained intermediates
x = img.clone().requires_grad_(True)
conv_in, conv_out = [], []
h = x
layout = O.discriminator_layout()
for j, (ci, _cin, _cout, stride, bi) in enumerate(layout):
    hq = O._bf16r(h); hq.retain_grad(); conv_in.append(hq)
    y = F.conv2d(hq, O._bf16r(sd[f"model.{ci}.weight"]), sd[f"model.{ci}.bias"], stride=stride, padding=1)
    y.retain_grad(); conv_out.append(y)
    if j == len(layout) - 1:
        h = y; break
    if bi is not None:
        y = F.batch_norm(y, None, None, sd[f"model.{bi}.weight"], sd[f"model.{bi}.bias"], training=True, eps=1e-5)
    h = y * torch.where(masks[j][:, :y.shape[1]], 1.0, 0.2) if PIN else F.leaky_relu(y, 0.2)
(h * r).sum().backward()
# --- HIP path with the backward's tensors captured
cap = {"dyb": [], "dx": []}
cb = cc._conv_backward
def conv_backward(x_, w, b, dyb, stride, need_dx, need_dw, **kw):
    out = cb(x_, w, b, dyb, stride, need_dx, need_dw, **kw)
    cap["dyb"].append((int(w.shape[0]), dyb.detach().clone(), kw.get("dy_split", False)))
    cap["dx"].append((int(w.shape[1]), None if out[0] is None else out[0].detach().clone()))
    return out
cc._conv_backward = conv_backward
netD = S.Discriminator(); netD.load_state_dict(sd); netD = netD.to(DEV).train()
xg = img.to(DEV).requires_grad_(True)
(netD(xg) * r.to(DEV)).sum().backward()
def blocked_to_nchw(t, C):
    if t.dtype == torch.float32:
        return cc.f32_as_nchw(t)[:, :C].float().cpu()
    return cc.blocked_as_nchw(t)[:, :C].float().cpu()
n = len(layout)
for idx in range(n):            # captured in reverse layer order
    j = n - 1 - idx
    cout, dyb, split = cap["dyb"][idx]
    g = blocked_to_nchw(dyb, 2 * cout if split else cout)
    if split: g = g[:, :cout] + g[:, cout:]
    cin, dx = cap["dx"][idx]
    e_out = rel(g, conv_out[j].grad)
    e_in = rel(blocked_to_nchw(dx, cin), conv_in[j].grad) if dx is not None else float("nan")
    print(f"conv {layout[j][0]:2d} ({layout[j][1]:3d}->{layout[j][2]:3d} s{layout[j][3]}): grad of its output {e_out:.4f}   grad of its input {e_in:.4f}")
