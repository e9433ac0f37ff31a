"""GPU: ``torch.library.opcheck`` on the registered ``sr355::`` ops — the declared schema (which arguments an op
mutates, what it aliases) against what the real kernels do, and the fake implementations against the real outputs."""
import pytest
import torch

import super_resolution_amd as S  # noqa: F401  (registers the ops)
from super_resolution_amd import ops
from super_resolution_amd._lib import DTYPE_BF16

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
UTILS = ("test_schema", "test_faketensor")


def test_opcheck_layout_pack_conv_adam():
    g = torch.Generator().manual_seed(3)
    x = torch.rand(2, 20, 9, 33, generator=g).to(DEV)
    torch.library.opcheck(torch.ops.sr355.to_blocked, (x, 8, DTYPE_BF16), test_utils=UTILS)
    b = ops.to_blocked(x, 8, DTYPE_BF16)
    torch.library.opcheck(torch.ops.sr355.from_blocked, (b, 0, 20), test_utils=UTILS)
    w, bias = (torch.rand(32, 20, 3, 3, generator=g) - 0.5).to(DEV), torch.rand(32, generator=g).to(DEV)
    torch.library.opcheck(torch.ops.sr355.pack_weight, (w, bias, DTYPE_BF16), test_utils=UTILS)
    wp, bp = ops.pack_weight(w, bias, DTYPE_BF16)
    # growth conv of a dense block: reads units [0, 4), writes 32 channels at unit 4 of the same buffer
    torch.library.opcheck(torch.ops.sr355.conv3x3_fwd,
                          (b, 4, wp, bp, 32, 0.2, 0, None, 0.0, None, 0.0, b, 4, None, None), test_utils=UTILS)
    out = torch.empty(2, 32, 9, 33, device=DEV)
    torch.library.opcheck(torch.ops.sr355.conv3x3_fwd,
                          (b, 4, wp, bp, 32, 1.0, 0, None, 0.0, None, 0.0, None, 0, None, out), test_utils=UTILS)
    dw, db = torch.empty(32, 20, 3, 3, device=DEV), torch.empty(32, device=DEV)
    torch.library.opcheck(torch.ops.sr355.conv3x3_wgrad, (b, 20, 0, b, 4, 32, 1.0, 0, dw, db), test_utils=UTILS)
    p, gr = torch.rand(1000, generator=g).to(DEV), torch.rand(1000, generator=g).to(DEV)
    m, v = torch.zeros(1000, device=DEV), torch.zeros(1000, device=DEV)
    torch.library.opcheck(torch.ops.sr355.adam_step, (p, gr, m, v, 1e-4, 0.9, 0.999, 1e-8, 0.0, 1), test_utils=UTILS)


def test_opcheck_generator_ops():
    net = S.GeneratorRRDB(num_blocks=1).to(DEV)
    cfg = net.cfg()
    weights = [c.weight.detach() for c in net.conv_modules()]
    biases = [c.bias.detach() for c in net.conv_modules()]
    torch.library.opcheck(torch.ops.sr355.generator_pack, (weights, biases, cfg), test_utils=UTILS)
    packed = ops.generator_pack(weights, biases, cfg)
    N, H, W = 1, 8, 12
    x = torch.rand(N, 3, H, W, generator=torch.Generator().manual_seed(4)).to(DEV)
    ws = torch.empty(ops.generator_workspace_bytes(cfg, N, H, W), dtype=torch.uint8, device=DEV)
    torch.library.opcheck(torch.ops.sr355.generator_forward, (x, packed, ws, cfg), test_utils=UTILS)
    saved = torch.empty(ops.generator_saved_bytes(cfg, N, H, W), dtype=torch.uint8, device=DEV)
    tws = torch.empty(ops.generator_train_workspace_bytes(cfg, N, H, W), dtype=torch.uint8, device=DEV)
    torch.library.opcheck(torch.ops.sr355.generator_forward_train, (x, packed, saved, tws, cfg), test_utils=UTILS)
    out = ops.generator_forward_train(x, packed, saved, tws, cfg)
    torch.library.opcheck(torch.ops.sr355.generator_pack_bwd, (weights, cfg), test_utils=UTILS)
    pb = ops.generator_pack_bwd(weights, cfg)
    bws = torch.empty(ops.generator_bwd_workspace_bytes(cfg, N, H, W), dtype=torch.uint8, device=DEV)
    torch.library.opcheck(torch.ops.sr355.generator_backward, (torch.ones_like(out), pb, saved, bws, cfg, H, W, True),
                          test_utils=UTILS)
