#!/usr/bin/env python3
"""Per-layer forward / data-gradient / weight-gradient times of the VGG19 feature stack and the discriminator at the
adversarial iteration's shapes (16 or 32 images of 128 x 128): which layers are far from the matrix peak.
python3 tools/time_layers.py [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import super_resolution_amd as S
from super_resolution_amd import conv_core as cc


def timed(fn, reps=10):
    """Device time of the kernels ``fn`` launches, per call, in microseconds (the host's launch path, which exceeds the
    kernels' time for the small layers, is not in it)."""
    from torch.profiler import ProfilerActivity, profile
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
    dev = [ev for ev in prof.events() if ev.device_type == torch.autograd.DeviceType.CUDA]
    return sum(ev.device_time_total if hasattr(ev, "device_time_total") else ev.cuda_time_total for ev in dev) / reps


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    vgg = S.VGGFeatureExtractor(pretrained=False).to(dev).eval()
    netD = S.Discriminator().to(dev)
    total = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    for name, convs in (("vgg", [(m, 1) for m in vgg.features if isinstance(m, torch.nn.Conv2d)]),
                        ("D", [(m, int(m.stride[0])) for m in netD.model if isinstance(m, torch.nn.Conv2d)])):
        h = w = 128
        pools = {2, 4, 8, 12} if name == "vgg" else set()
        for i, (m, stride) in enumerate(convs):
            cin, cout = m.in_channels, m.out_channels
            if cout % 4:
                continue
            x = cc.to_blocked(torch.randn(batch, cin, h, w, device=dev))
            wgt, bias = m.weight.detach(), m.bias.detach()
            ho, wo = ((h + 1) // 2, (w + 1) // 2) if stride == 2 else (h, w)
            with torch.no_grad():
                f = timed(lambda: cc._conv_forward(x, wgt, bias, 0.2, "blocked", stride))
                dy = cc.to_blocked(torch.randn(batch, cout, ho, wo, device=dev))
                d = timed(lambda: cc._conv_backward(x, wgt, bias, dy, stride, True, False))
                g = timed(lambda: cc._conv_backward(x, wgt, bias, dy, stride, False, True))
            flop = 2.0 * 9 * cin * cout * batch * ho * wo
            print(f"{name}{i:2d} {cin:4d}->{cout:4d} s{stride} {h:3d}x{w:<3d}: fwd {f:7.1f} us {flop / f / 1e6:6.0f} TF/s | "
                  f"dgrad {d:7.1f} us {flop / d / 1e6:6.0f} | wgrad {g:7.1f} us {flop / g / 1e6:6.0f}")
            total["fwd"] += f
            total["dgrad"] += d
            total["wgrad"] += g
            h, w = ho, wo
            if (i + 1) in pools:
                h, w = h // 2, w // 2
    print("sum (us):", {k: round(v) for k, v in total.items()})


if __name__ == "__main__":
    main()
