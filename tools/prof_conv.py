#!/usr/bin/env python3
"""Launch each distinct body-conv configuration of BASELINE config 2 a few times (for rocprofv3 passes).

    python3 tools/prof_conv.py [reps]            # env SR355_RING=0/1 selects the kernel variant
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import super_resolution_amd as S
from super_resolution_amd import _lib, ops


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = S.GeneratorRRDB(num_blocks=1).to(dev)
    N, H, W, nf, gc = 16, 128, 128, 64, 32
    dt = S.DTYPE_BF16
    rdb = net.body[0].rdb1
    packed = [ops.pack_weight(c.weight.detach(), c.bias.detach(), dt) for c in rdb._convs()]
    buf = torch.zeros((N, H, 24, W, 8), dtype=torch.bfloat16, device=dev).normal_()
    nxt = torch.empty_like(buf)
    trunk = torch.zeros((N, H, 16, W, 4), dtype=torch.float32, device=dev).normal_()
    L = _lib.lib()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def args(cin_units, wp, bp, cout, slope, out_act, off, res1=None, out_f32=None):
        a = _lib.ConvArgs()
        a.dtype, a.N, a.H, a.W, a.upsample = dt, N, H, W, 0
        a.in_dev, a.in_bytes, a.in_units, a.cin_units = buf.data_ptr(), buf.numel() * 2, 24, cin_units
        a.w_packed_dev, a.w_packed_bytes, a.weight_split = wp.data_ptr(), wp.numel(), 0
        a.bias_packed_dev, a.cout, a.lrelu_slope = bp.data_ptr(), cout, slope
        if res1 is not None:
            a.res1_dev, a.res1_weight, a.res1_scale, a.res_bytes = res1.data_ptr(), 1.0, 0.2, res1.numel() * 4
        a.out_act_dev, a.out_act_bytes, a.out_units, a.out_unit_off = out_act.data_ptr(), out_act.numel() * 2, 24, off
        if out_f32 is not None:
            a.out_f32_dev = out_f32.data_ptr()
        return a

    calls = []
    for k in range(4):
        wp, bp = packed[k]
        calls.append(args(8 + 4 * k, wp, bp, gc, 0.2, buf, 8 + 4 * k))
    wp, bp = packed[4]
    calls.append(args(24, wp, bp, nf, 1.0, nxt, 0, res1=trunk, out_f32=trunk))
    for _ in range(reps):
        for a in calls:
            rc = L.sr355_conv3x3_fwd(C.byref(a), stream)
            assert rc == 0, L.sr355_last_error()
    torch.cuda.synchronize()
    print("done", reps)


if __name__ == "__main__":
    main()
