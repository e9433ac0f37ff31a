"""CPU: the built code objects hold no wide store whose data registers the next instructions overwrite.

A vector-memory store of more than 64 bits reads its data registers for two wait states after it issues.  hipcc inserts
them unless the store's scalar offset is an SGPR (or the store sits in an inline-asm statement); gfx950 needs them
either way.  The discriminator's non-reproducible forward pass came from exactly such a pair in the conv kernels' fp32
epilogue (csrc/conv_epilogue.h, epi_store_b128).  This test disassembles the gfx950 code objects inside libsr355.so
and looks for the pattern, so a future store written the hazardous way fails here, on the CPU, instead of one run in a
hundred on the GPU."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_store_hazard as H  # noqa: E402

LIB = os.path.join(ROOT, "super_resolution_amd", "libsr355.so")


@pytest.mark.skipif(not os.path.exists(os.path.join(H.LLVM, "llvm-objdump")), reason="no llvm-objdump in this image")
def test_no_wide_store_is_overwritten_inside_its_wait_states():
    assert os.path.exists(LIB), "libsr355.so is not built (python3 -c 'import __graft_entry__ as g; g.build()')"
    total, bad = H.check_library(LIB)
    assert total > 500, f"only {total} wide stores found: the disassembly did not see the kernels"
    assert not bad, "\n".join(f"{k}: {line}  <- {nxt} after {ws} wait state(s)" for _n, k, _no, line, ws, nxt in bad)


def test_the_scanner_sees_the_hazard_it_is_for():
    text = """
_Z4kern:
	buffer_store_dwordx4 v[72:75], v88, s[96:99], s27 offen
	v_add_f32_e32 v72, v80, v40
	buffer_store_dwordx4 v[72:75], v88, s[96:99], 0 offen
	s_nop 1
	v_add_f32_e32 v72, v80, v40
	global_store_dwordx4 v[2:3], v[10:13], off
	v_mov_b32_e32 v9, 0
	v_mov_b32_e32 v13, 0
	buffer_store_dwordx2 v[72:73], v88, s[96:99], s27 offen
	v_add_f32_e32 v72, v80, v40
"""
    total, bad = H.scan(text, "sample")
    assert total == 3
    assert [(b[4], b[5]) for b in bad] == [(0, "v_add_f32_e32 v72, v80, v40"), (1, "v_mov_b32_e32 v13, 0")]
