"""GPU: the one-launch dense block (``sr355_dense_block_fwd``) against the five ``sr355_conv3x3_fwd`` launches it
replaces (/root/reference/esrgan/models/esrgan_model.py:18-24), in its forward form (bias, LeakyReLU, fp32 skip
carriers) and in its data-gradient form (transposed packs, zero bias, sign masks).  Both paths use the same MFMA
tiles in the same accumulation order and round at the same points, so every output must be bit-identical; the
chain is additionally checked against a torch fp32 restatement of the block within bf16 tolerance."""
import pytest
import torch
import torch.nn.functional as F

from super_resolution_amd import ops
from super_resolution_amd._lib import DTYPE_BF16, DTYPE_F32

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NF, GC = 64, 32

SHAPES = [(2, 32, 32), (1, 7, 19), (3, 1, 32), (1, 33, 5), (16, 32, 32), (1, 2, 1)]


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def make_convs(seed, scale=0.03):
    ws = [rnd(GC if k < 4 else NF, NF + k * GC, 3, 3, seed=seed + k, scale=scale) for k in range(5)]
    bs = [rnd(GC if k < 4 else NF, seed=seed + 10 + k, scale=0.1) for k in range(5)]
    return ws, bs


def five_launches(buf, packs, biases, slope, res1, s1, res2, s2, out_act, out_f32, mask=None, mask_off=None,
                  mask_slope=1.0, r1w=1.0, r2w=1.0, res_act=None):
    for k in range(4):
        ops.conv3x3_fwd(buf, 8 + 4 * k, packs[k], biases[k], GC, slope, 0, None, 0.0, None, 0.0, buf, 8 + 4 * k,
                        None, None, mask=mask, mask_unit_off=(mask_off[k] if mask is not None else 0),
                        mask_slope=mask_slope)
    a = ops.conv_args(buf, 24, packs[4], biases[4], NF, 1.0, 0, res1, s1, res2, s2, out_act, 0, out_f32, None,
                      res1_weight=r1w, res2_weight=r2w, res_act=res_act)
    ops.launch_conv(a)


@pytest.mark.parametrize("shape", SHAPES)
def test_forward_form_is_bit_identical_to_five_launches(shape):
    N, H, W = shape
    ws, bs = make_convs(100)
    packed = [ops.pack_weight(w.to(DEV), b.to(DEV), DTYPE_BF16) for w, b in zip(ws, bs)]
    packs, biases = [p[0] for p in packed], [p[1] for p in packed]
    x = rnd(N, NF, H, W, seed=1)
    t_in, r_in = rnd(N, NF, H, W, seed=2), rnd(N, NF, H, W, seed=3)
    res1 = ops.to_blocked(t_in.to(DEV), NF // 4, DTYPE_F32)
    res2 = ops.to_blocked(r_in.to(DEV), NF // 4, DTYPE_F32)
    outs = []
    for fused in (False, True):
        buf = ops.to_blocked(x.to(DEV), 24, DTYPE_BF16)
        buf[:, :, 8:] = 7.0           # stale growth channels must not matter
        nxt = torch.full_like(buf, 3.0)
        out_f32 = torch.zeros((N, H, NF // 4, W, 4), dtype=torch.float32, device=DEV)
        if fused:
            ops.dense_block_fwd(buf, packs, biases, 0.2, True, res1, 0.2, res2, 0.2, nxt, out_f32)
        else:
            five_launches(buf, packs, biases, 0.2, res1, 0.2, res2, 0.2, nxt, out_f32)
        torch.cuda.synchronize()
        outs.append((buf.clone(), nxt[:, :, :8].clone(), out_f32.clone()))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    # and the chain is the reference's dense block (esrgan_model.py:18-24 restated here with bf16-rounded operands)
    bf = lambda t: t.to(torch.bfloat16).float()
    x1 = bf(x)
    feats = [x1]
    for k in range(4):
        feats.append(F.leaky_relu(F.conv2d(torch.cat(feats, 1), bf(ws[k]), bs[k], padding=1), 0.2))
    x5 = F.conv2d(torch.cat(feats, 1), bf(ws[4]), bs[4], padding=1)
    ref = r_in + 0.2 * (t_in + 0.2 * x5)     # the RRDB skip folded over the dense skip (:24, :39)
    got = ops.from_blocked(outs[1][2], 0, NF).cpu()
    assert (got - ref).abs().max().item() < 3e-2 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("shape", SHAPES[:4])
def test_growth_outputs_can_stay_in_lds_and_skip_can_read_the_activations(shape):
    """Inference form: growth outputs not stored, dense skip read from the bf16 block input (no fp32 carrier)."""
    N, H, W = shape
    ws, bs = make_convs(200)
    packed = [ops.pack_weight(w.to(DEV), b.to(DEV), DTYPE_BF16) for w, b in zip(ws, bs)]
    packs, biases = [p[0] for p in packed], [p[1] for p in packed]
    x = rnd(N, NF, H, W, seed=4)
    outs = []
    for fused in (False, True):
        buf = ops.to_blocked(x.to(DEV), 24, DTYPE_BF16)
        buf[:, :, 8:] = 7.0
        nxt = torch.full_like(buf, 3.0)
        if fused:
            ops.dense_block_fwd(buf, packs, biases, 0.2, False, None, 0.2, None, 0.0, nxt, None, res_act=buf)
            torch.cuda.synchronize()
            assert (buf[:, :, 8:] == 7.0).all()      # untouched
        else:
            five_launches(buf, packs, biases, 0.2, None, 0.2, None, 0.0, nxt, None, res_act=buf)
        torch.cuda.synchronize()
        outs.append(nxt[:, :, :8].clone())
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("shape", SHAPES)
def test_data_gradient_form_is_bit_identical_to_five_launches(shape):
    """Backward of a dense block: the stack [dOut | dY4 | dY3 | dY2 | dY1] grows through four masked convs, the
    fifth adds weighted fp32 carriers (what api.hip's backward schedule issues per dense block)."""
    N, H, W = shape
    ws, _ = make_convs(300)
    # transposed packs of the forward weights, stacked as the backward schedule stacks them
    s5 = 0.2
    packs = []
    for j in (4, 3, 2, 1, 0):
        ci_lo, n_out = (NF + (j - 1) * GC, GC) if j else (0, NF)
        units = (NF + (4 - j) * GC) // 8
        wt = torch.zeros(units * 9 * ((n_out + 31) // 32 * 32) * 16, dtype=torch.uint8, device=DEV)
        ops.pack_weight_transposed(ws[4].to(DEV), ci_lo, n_out, s5, wt, units, 0, DTYPE_BF16)
        for k in range(4, j, -1):
            ops.pack_weight_transposed(ws[k - 1].to(DEV), ci_lo, n_out, 1.0, wt, units, (NF + (4 - k) * GC) // 8,
                                       DTYPE_BF16)
        packs.append(wt)
    zero = torch.zeros(64, dtype=torch.float32, device=DEV)
    biases = [zero] * 5
    act = ops.to_blocked(rnd(N, NF + 4 * GC, H, W, seed=5).to(DEV), 24, DTYPE_BF16)   # saved forward activations
    mask_off = [20, 16, 12, 8]
    dout = rnd(N, NF, H, W, seed=6)
    gT = ops.to_blocked(rnd(N, NF, H, W, seed=7).to(DEV), NF // 4, DTYPE_F32)
    gR = ops.to_blocked(rnd(N, NF, H, W, seed=8).to(DEV), NF // 4, DTYPE_F32)
    outs = []
    for fused in (False, True):
        stack = ops.to_blocked(dout.to(DEV), 24, DTYPE_BF16)
        stack[:, :, 8:] = 5.0
        nxt = torch.full_like(stack, 3.0)
        out_f32 = torch.zeros_like(gT)
        if fused:
            ops.dense_block_fwd(stack, packs, biases, 1.0, True, gT, 1.0, gR, 1.0, nxt, out_f32, mask=act,
                                mask_unit_off=mask_off, mask_slope=0.2, res1_weight=1.0, res2_weight=0.2)
        else:
            five_launches(stack, packs, biases, 1.0, gT, 1.0, gR, 1.0, nxt, out_f32, mask=act, mask_off=mask_off,
                          mask_slope=0.2, r1w=1.0, r2w=0.2)
        torch.cuda.synchronize()
        outs.append((stack.clone(), nxt[:, :, :8].clone(), out_f32.clone()))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    assert outs[1][2].abs().max().item() > 0


def test_wider_images_and_bad_arguments_are_refused():
    ws, bs = make_convs(400)
    packed = [ops.pack_weight(w.to(DEV), b.to(DEV), DTYPE_BF16) for w, b in zip(ws, bs)]
    packs, biases = [p[0] for p in packed], [p[1] for p in packed]
    buf = torch.zeros((1, 4, 24, 33, 8), dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RuntimeError, match="32 pixels"):
        ops.dense_block_fwd(buf, packs, biases, 0.2, True, None, 0.0, None, 0.0, torch.empty_like(buf), None)
    buf = torch.zeros((1, 4, 24, 16, 8), dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RuntimeError, match="no output"):
        ops.dense_block_fwd(buf, packs, biases, 0.2, True, None, 0.0, None, 0.0, None, None)
    with pytest.raises(RuntimeError, match="conv3"):
        ops.dense_block_fwd(buf, packs[:2] + [packs[0]] + packs[3:], biases, 0.2, True, None, 0.0, None, 0.0,
                            torch.empty_like(buf), None)
