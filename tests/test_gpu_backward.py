"""GPU: backward-pass primitives (data gradient via transposed packing, weight gradient, 2x2 sum
pooling) against torch autograd on the CPU.  bf16 storage, fp32 accumulation: inputs are rounded to
bf16 on both sides, so tolerances only cover summation order and the final bf16 store."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import super_resolution_amd as S
from super_resolution_amd import _lib, ops
from super_resolution_amd._lib import DTYPE_BF16, DTYPE_F32

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def bf16r(x):
    return x.to(torch.bfloat16).to(torch.float32)


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def packed_zero_bias(n):
    return torch.zeros((n + 31) // 32 * 32, dtype=torch.float32, device=DEV)


@pytest.mark.parametrize("dtype", [DTYPE_BF16, DTYPE_F32])
@pytest.mark.parametrize("case", [(1, 64, 32, 9, 37), (2, 96, 32, 16, 32), (1, 192, 64, 5, 33)])
def test_data_gradient_is_the_forward_kernel_on_transposed_weights(dtype, case):
    N, cin, cout, H, W = case
    cpu = 8 if dtype == DTYPE_BF16 else 4
    w = rnd(cout, cin, 3, 3, seed=1, scale=0.05)
    dy = rnd(N, cout, H, W, seed=2)
    if dtype == DTYPE_BF16:
        w_ref, dy_ref = bf16r(w), bf16r(dy)
    else:
        w_ref, dy_ref = w, dy
    ref = F.conv_transpose2d(dy_ref.double(), w_ref.double(), padding=1).float()   # = d(conv2d)/d(input)
    units = (cout + cpu - 1) // cpu
    units += units & 1
    wt = torch.zeros(units * 9 * ((cin + 31) // 32 * 32) * 16, dtype=torch.uint8, device=DEV)
    ops.pack_weight_transposed(w.to(DEV), 0, cin, 1.0, wt, units, 0, dtype)
    dyb = ops.to_blocked(dy.to(DEV), units, dtype)
    out = torch.zeros((N, cin, H, W), dtype=torch.float32, device=DEV)
    # cout of the dgrad conv = cin of the forward conv; wider than 64 -> several output-channel blocks
    ops.conv3x3_fwd(dyb, units, wt, packed_zero_bias(cin), cin, 1.0, 0, None, 0.0, None, 0.0, None, 0, None, out)
    torch.cuda.synchronize()
    assert (out.cpu() - ref).abs().max().item() < 2e-4


def test_data_gradient_channel_slices_stack_like_a_dense_block():
    """d(x) of a 2-conv stack: conv_b reads [x, a] and conv_a reads x; the gradient wrt x's 64 channels is
    ONE conv over the stacked output gradients [dYb, dYa] with the stacked transposed weight slices."""
    dtype, cpu = DTYPE_BF16, 8
    N, H, W = 1, 10, 34
    wa = rnd(32, 64, 3, 3, seed=3, scale=0.05)
    wb = rnd(64, 96, 3, 3, seed=4, scale=0.05)
    dya, dyb = rnd(N, 32, H, W, seed=5), rnd(N, 64, H, W, seed=6)
    ref = (F.conv_transpose2d(bf16r(dyb).double(), bf16r(wb)[:, :64].double(), padding=1)
           + F.conv_transpose2d(bf16r(dya).double(), bf16r(wa).double(), padding=1)).float()
    units = 12                                            # 64 (dYb) + 32 (dYa) channels
    wt = torch.zeros(units * 9 * 64 * 16, dtype=torch.uint8, device=DEV)
    ops.pack_weight_transposed(wb.to(DEV), 0, 64, 1.0, wt, units, 0, dtype)
    ops.pack_weight_transposed(wa.to(DEV), 0, 64, 1.0, wt, units, 8, dtype)
    stack = ops.to_blocked(torch.cat([dyb, dya], 1).to(DEV), units, dtype)
    out = torch.zeros((N, 64, H, W), dtype=torch.float32, device=DEV)
    ops.conv3x3_fwd(stack, units, wt, packed_zero_bias(64), 64, 1.0, 0, None, 0.0, None, 0.0, None, 0, None, out)
    torch.cuda.synchronize()
    assert (out.cpu() - ref).abs().max().item() < 3e-4


def test_epilogue_mask_and_weighted_residuals():
    """v = res1_w*res1 + s1*(mask > 0 ? conv : slope*conv): LeakyReLU backward plus the skip gradient."""
    dtype, cpu = DTYPE_BF16, 8
    N, cin, cout, H, W = 1, 64, 64, 12, 35
    x, w = rnd(N, cin, H, W, seed=7), rnd(cout, cin, 3, 3, seed=8, scale=0.05)
    act = rnd(N, cout, H, W, seed=9)
    res = rnd(N, cout, H, W, seed=10)
    conv = F.conv2d(bf16r(x).double(), bf16r(w).double(), padding=1)
    ref = (0.2 * res.double() + 1.5 * torch.where(bf16r(act).double() > 0, conv, 0.2 * conv)).float()
    xb = ops.to_blocked(x.to(DEV), 8, dtype)
    wp, bp = ops.pack_weight(w.to(DEV), torch.zeros(cout, device=DEV), dtype)
    mask = ops.to_blocked(torch.cat([torch.zeros(N, 16, H, W), act], 1).to(DEV), 10, dtype)   # act at unit 2
    r1 = ops.to_blocked(res.to(DEV), cout // 4, DTYPE_F32)
    out = torch.zeros((N, H, cout // 4, W, 4), dtype=torch.float32, device=DEV)
    ops.conv3x3_fwd(xb, 8, wp, bp, cout, 1.0, 0, r1, 1.5, None, 0.0, None, 0, out, None,
                    mask=mask, mask_unit_off=2, mask_slope=0.2, res1_weight=0.2)
    torch.cuda.synchronize()
    assert (ops.from_blocked(out, 0, cout).cpu() - ref).abs().max().item() < 2e-4


@pytest.mark.parametrize("dtype", [DTYPE_BF16, DTYPE_F32])
def test_pool2x2_sum_is_the_adjoint_of_nearest_upsampling(dtype):
    N, C, H, W = 2, 64, 7, 19
    g = rnd(N, C, 2 * H, 2 * W, seed=11)
    act = rnd(N, C, H, W, seed=12)
    gq = bf16r(g) if dtype == DTYPE_BF16 else g
    aq = bf16r(act) if dtype == DTYPE_BF16 else act
    ref = F.avg_pool2d(gq.double(), 2) * 4
    ref = torch.where(aq.double() > 0, ref, 0.2 * ref).float()
    cpu = 8 if dtype == DTYPE_BF16 else 4
    gb = ops.to_blocked(g.to(DEV), C // cpu, dtype)
    mask = ops.to_blocked(act.to(DEV), C // cpu, dtype)
    out_f32 = torch.zeros((N, H, C // 4, W, 4), dtype=torch.float32, device=DEV)
    out_act = torch.zeros((N, H, C // cpu + 2, W, cpu), dtype=gb.dtype, device=DEV)
    ops.pool2x2_sum(gb, mask, 0, 0.2, out_f32, out_act, 2)
    torch.cuda.synchronize()
    got = ops.from_blocked(out_f32, 0, C).cpu()
    assert (got - ref).abs().max().item() < 1e-5
    got_act = ops.from_blocked(out_act, 2, C).cpu()
    assert torch.equal(got_act, bf16r(got) if dtype == DTYPE_BF16 else got)


WGRAD_CASES = [
    # (N, cin, cout, H, W, upsample)
    (1, 64, 32, 8, 32, 0),
    (2, 96, 32, 5, 37, 0),       # ragged tile edges
    (1, 192, 64, 9, 33, 0),
    (3, 64, 64, 16, 64, 0),      # several tiles per workgroup
    (1, 64, 64, 14, 38, 1),      # input stored at half resolution (nearest x2 fused in the forward)
    (1, 3, 64, 6, 40, 0),        # conv_first: 3 input channels in a padded unit
    (1, 64, 3, 9, 35, 0),        # conv_last: 3 output channels
    (2, 40, 96, 7, 33, 0),       # more than 64 outputs: 32-channel slices as the jobs of one launch
    (1, 64, 160, 12, 16, 0),
    (1, 16, 544, 6, 20, 0),      # 17 slices: two launches
]


@pytest.mark.parametrize("case", WGRAD_CASES)
def test_weight_and_bias_gradient(case):
    N, cin, cout, H, W, ups = case
    Hin, Win = H >> ups, W >> ups
    x = rnd(N, cin, Hin, Win, seed=20)
    dy = rnd(N, cout, H, W, seed=21)
    xq, dyq = bf16r(x).double(), bf16r(dy).double()
    xin = F.interpolate(xq, scale_factor=2, mode="nearest") if ups else xq
    w0 = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    b0 = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
    (F.conv2d(xin, w0, b0, padding=1) * dyq).sum().backward()
    in_units = (cin + 7) // 8 + 1
    xb = ops.to_blocked(x.to(DEV), in_units, DTYPE_BF16)
    dy_units = (cout + 7) // 8 + 3
    dyb = torch.zeros((N, H, dy_units, W, 8), dtype=torch.bfloat16, device=DEV)
    dyb[:, :, 3:] = ops.to_blocked(dy.to(DEV), (cout + 7) // 8, DTYPE_BF16)
    dw = torch.full((cout, cin, 3, 3), 7.0, dtype=torch.float32, device=DEV)
    db = torch.full((cout,), 7.0, dtype=torch.float32, device=DEV)
    ops.conv3x3_wgrad(xb, cin, ups, dyb, 3, cout, 0.5, 0, dw, db)
    torch.cuda.synchronize()
    scale_w = w0.grad.abs().max().item()
    assert (dw.cpu().double() - 0.5 * w0.grad).abs().max().item() < 1e-4 * max(1.0, scale_w)
    assert (db.cpu().double() - 0.5 * b0.grad).abs().max().item() < 1e-4 * max(1.0, b0.grad.abs().max().item())
    # accumulate mode adds on top
    ops.conv3x3_wgrad(xb, cin, ups, dyb, 3, cout, 0.5, 1, dw, db)
    torch.cuda.synchronize()
    assert (dw.cpu().double() - w0.grad).abs().max().item() < 2e-4 * max(1.0, scale_w)


@pytest.mark.parametrize("shape", [(16, 32, 32), (2, 9, 37), (1, 64, 96)])
def test_dense_block_weight_gradients_in_one_launch_equal_five_launches(shape):
    """sr355_dense_block_wgrad (the five weight / bias gradients of a dense block, autograd of
    /root/reference/esrgan/models/esrgan_model.py:18-24, as one batched launch + one reduction) against five
    sr355_conv3x3_wgrad calls on the same buffers: the same kernels and summation order, so bit-identical."""
    N, H, W = shape
    nf, gc, units = 64, 32, 24
    act = torch.randn((N, H, units, W, 8), generator=torch.Generator().manual_seed(30)).to(DEV).to(torch.bfloat16)
    stack = torch.randn((N, H, units, W, 8), generator=torch.Generator().manual_seed(31)).to(DEV).to(torch.bfloat16)
    dws = [torch.full((gc if k < 4 else nf, nf + k * gc, 3, 3), 3.0, dtype=torch.float32, device=DEV) for k in range(5)]
    dbs = [torch.full((gc if k < 4 else nf,), 3.0, dtype=torch.float32, device=DEV) for k in range(5)]
    ops.dense_block_wgrad(act, stack, nf, gc, 0.7, 0.14, dws, dbs)
    for k in range(5):
        cin, cout = nf + k * gc, (gc if k < 4 else nf)
        dw = torch.empty((cout, cin, 3, 3), dtype=torch.float32, device=DEV)
        db = torch.empty((cout,), dtype=torch.float32, device=DEV)
        dy_off = (nf + (3 - k) * gc) // 8 if k < 4 else 0
        ops.conv3x3_wgrad(act, cin, 0, stack, dy_off, cout, 0.7 if k < 4 else 0.14, 0, dw, db)
        torch.cuda.synchronize()
        assert torch.equal(dw, dws[k]) and torch.equal(db, dbs[k]), k
    with pytest.raises(RuntimeError):
        ops.dense_block_wgrad(act, stack[:, :, :20].contiguous(), nf, gc, 1.0, 0.2, dws, dbs)


@pytest.mark.gpu
@pytest.mark.parametrize("n_blocks", [2, 3, 6, 9])
def test_several_dense_blocks_in_one_weight_gradient_launch_equal_single_launches(n_blocks):
    """sr355_dense_blocks_wgrad: the weight gradients of the dense blocks of one or two RRDBs
    (/root/reference/esrgan/models/esrgan_model.py:30-33) as one launch + one reduction, through the C ABI, against one
    sr355_dense_block_wgrad call per block: every block keeps its own slabs and summation order, so bit-identical.  Seven
    blocks, differing shapes and a scratch buffer of one block's size are refused."""
    from super_resolution_amd import _lib
    N, H, W, nf, gc, units = 16, 32, 32, 64, 32, 24
    L = _lib.lib()
    one = L.sr355_dense_block_wgrad_workspace_bytes(nf, gc, N, H, W)
    ws = torch.empty(n_blocks * one, dtype=torch.uint8, device=DEV)
    gen = torch.Generator().manual_seed(40 + n_blocks)
    blocks, expect = [], []
    keep = []
    for b in range(n_blocks):
        act = torch.randn((N, H, units, W, 8), generator=gen).to(DEV).to(torch.bfloat16)
        stack = torch.randn((N, H, units, W, 8), generator=gen).to(DEV).to(torch.bfloat16)
        dws = [torch.full((gc if k < 4 else nf, nf + k * gc, 3, 3), 5.0, dtype=torch.float32, device=DEV) for k in range(5)]
        dbs = [torch.full((gc if k < 4 else nf,), 5.0, dtype=torch.float32, device=DEV) for k in range(5)]
        s5 = 0.04 if b % 3 == 0 else 0.2
        blocks.append(ops.dense_wgrad_args(act, stack, nf, gc, 0.7, s5, dws, dbs, ws))
        rdw = [torch.empty_like(t) for t in dws]
        rdb = [torch.empty_like(t) for t in dbs]
        ops.dense_block_wgrad(act, stack, nf, gc, 0.7, s5, rdw, rdb)
        expect.append((rdw, rdb))
        keep.append((act, stack, dws, dbs))
    ops.launch_dense_wgrad_blocks(blocks)
    torch.cuda.synchronize()
    for b in range(n_blocks):
        for k in range(5):
            assert torch.equal(keep[b][2][k], expect[b][0][k]) and torch.equal(keep[b][3][k], expect[b][1][k]), (b, k)
    with pytest.raises(RuntimeError):
        ops.launch_dense_wgrad_blocks(blocks * 7)
    if n_blocks > 1:
        small = torch.empty(one, dtype=torch.uint8, device=DEV)
        bad = list(blocks)
        bad[0] = ops.dense_wgrad_args(keep[0][0], keep[0][1], nf, gc, 0.7, 0.2, keep[0][2], keep[0][3], small)
        with pytest.raises(_lib.Sr355Error, match="scratch"):
            ops.launch_dense_wgrad_blocks(bad)
        other = torch.zeros((N, H // 2, units, W, 8), dtype=torch.bfloat16, device=DEV)
        bad = list(blocks)
        bad[1] = ops.dense_wgrad_args(other, other.clone(), nf, gc, 0.7, 0.2, keep[1][2], keep[1][3], ws)
        with pytest.raises(_lib.Sr355Error, match="differs"):
            ops.launch_dense_wgrad_blocks(bad)
