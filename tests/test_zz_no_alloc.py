"""GPU: the allocation-free weight gradient ``sr355_conv3x3_wgrad_static`` (partial slabs in the library's static
device array) against ``sr355_conv3x3_wgrad`` (caller workspace), through the C ABI.  Both run the same kernels
in the same summation order, so the results must be bit-identical.  Two calls in a row share the array: the second
is smaller and must not see what the first left behind."""
import ctypes as C

import pytest
import torch

from super_resolution_amd import _lib
from super_resolution_amd._lib import WgradArgs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _wgrad(entry, N, cin, cout, H, W, seed, with_workspace):
    L = _lib.lib()
    g = torch.Generator().manual_seed(seed)
    in_units = (cin + 7) // 8
    dy_units = (cout + 7) // 8 + 2
    x = torch.randn((N, H, in_units, W, 8), generator=g).to(torch.bfloat16).to(DEV)
    dy = torch.randn((N, H, dy_units, W, 8), generator=g).to(torch.bfloat16).to(DEV)
    dw = torch.full((cout, cin, 3, 3), 3.0, dtype=torch.float32, device=DEV)
    db = torch.full((cout,), 3.0, dtype=torch.float32, device=DEV)
    a = WgradArgs()
    a.N, a.H, a.W, a.upsample = N, H, W, 0
    a.in_dev, a.in_bytes, a.in_units, a.cin = x.data_ptr(), x.numel() * 2, in_units, cin
    a.dy_dev, a.dy_bytes, a.dy_units, a.dy_unit_off = dy.data_ptr(), dy.numel() * 2, dy_units, 2
    a.cout, a.scale, a.accumulate = cout, 0.75, 0
    a.dw_dev, a.dw_bytes, a.db_dev = dw.data_ptr(), dw.numel() * 4, db.data_ptr()
    ws = None
    if with_workspace:
        need = L.sr355_wgrad_workspace_bytes(cout, cin, N, H, W)
        ws = torch.empty(need, dtype=torch.uint8, device=DEV)
        a.workspace_dev, a.workspace_bytes = ws.data_ptr(), need
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(getattr(L, entry)(C.byref(a), stream), entry)
    torch.cuda.synchronize()
    return dw.cpu(), db.cpu()


def test_static_scratch_weight_gradient_matches_the_workspace_one_twice_in_a_row():
    # first call: a dense block's conv5 at the training shape (14 MB of slabs); second: a much smaller conv
    for (N, cin, cout, H, W, seed) in [(16, 192, 64, 32, 32, 1), (1, 64, 32, 9, 37, 2)]:
        ref_dw, ref_db = _wgrad("sr355_conv3x3_wgrad", N, cin, cout, H, W, seed, True)
        got_dw, got_db = _wgrad("sr355_conv3x3_wgrad_static", N, cin, cout, H, W, seed, False)
        assert torch.equal(got_dw, ref_dw)
        assert torch.equal(got_db, ref_db)
        assert ref_dw.abs().max().item() > 0


def test_static_scratch_call_allocates_no_device_memory():
    """After one warm-up call on the device (the first call takes the array's address and sets the kernel's LDS
    attribute), a call changes neither the driver's free memory nor torch's allocator counters: buffers are built
    beforehand, the entry point is called directly."""
    L = _lib.lib()
    N, cin, cout, H, W = 16, 160, 32, 32, 32
    g = torch.Generator().manual_seed(3)
    x = torch.randn((N, H, 20, W, 8), generator=g).to(torch.bfloat16).to(DEV)
    dy = torch.randn((N, H, 4, W, 8), generator=g).to(torch.bfloat16).to(DEV)
    dw = torch.empty((cout, cin, 3, 3), dtype=torch.float32, device=DEV)
    db = torch.empty((cout,), dtype=torch.float32, device=DEV)
    a = WgradArgs()
    a.N, a.H, a.W, a.upsample = N, H, W, 0
    a.in_dev, a.in_bytes, a.in_units, a.cin = x.data_ptr(), x.numel() * 2, 20, cin
    a.dy_dev, a.dy_bytes, a.dy_units, a.dy_unit_off = dy.data_ptr(), dy.numel() * 2, 4, 0
    a.cout, a.scale, a.accumulate = cout, 1.0, 0
    a.dw_dev, a.dw_bytes, a.db_dev = dw.data_ptr(), dw.numel() * 4, db.data_ptr()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(L.sr355_conv3x3_wgrad_static(C.byref(a), stream), "warm-up")
    torch.cuda.synchronize()
    free0, _total = torch.cuda.mem_get_info()
    stats0 = torch.cuda.memory_stats()
    for _ in range(3):
        _lib.check(L.sr355_conv3x3_wgrad_static(C.byref(a), stream), "sr355_conv3x3_wgrad_static")
    torch.cuda.synchronize()
    free1, _total = torch.cuda.mem_get_info()
    stats1 = torch.cuda.memory_stats()
    assert free1 == free0
    for key in ("allocation.all.allocated", "segment.all.allocated", "allocated_bytes.all.current",
                "reserved_bytes.all.current", "num_device_alloc"):
        assert stats1.get(key) == stats0.get(key), key


def test_static_scratch_rejects_shapes_beyond_its_size():
    L = _lib.lib()
    N, cin, cout, H, W = 64, 192, 64, 64, 64
    assert L.sr355_wgrad_workspace_bytes(cout, cin, N, H, W) > _lib.STATIC_SCRATCH_BYTES
    x = torch.zeros((N, H, 24, W, 8), dtype=torch.bfloat16, device=DEV)
    dw = torch.zeros((cout, cin, 3, 3), dtype=torch.float32, device=DEV)
    a = WgradArgs()
    a.N, a.H, a.W = N, H, W
    a.in_dev, a.in_bytes, a.in_units, a.cin = x.data_ptr(), x.numel() * 2, 24, cin
    a.dy_dev, a.dy_bytes, a.dy_units, a.dy_unit_off = x.data_ptr(), x.numel() * 2, 24, 0
    a.cout, a.scale = cout, 1.0
    a.dw_dev, a.dw_bytes = dw.data_ptr(), dw.numel() * 4
    rc = L.sr355_conv3x3_wgrad_static(C.byref(a), None)
    assert rc == -1
    assert b"scratch" in L.sr355_last_error()
