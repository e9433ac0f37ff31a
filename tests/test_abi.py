"""CPU: the C-ABI library loads, exports every symbol include/sr355.h declares, and its host-side
size/validation entry points behave (no kernel is launched here)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT
from super_resolution_amd import _lib
from super_resolution_amd._lib import ConvArgs, GeneratorCfg


def header_symbols():
    text = open(os.path.join(ROOT, "include", "sr355.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sr355_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def L():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.lib()


def test_header_and_binding_list_agree():
    assert header_symbols() == sorted(_lib.EXPORTS)


def test_every_declared_symbol_is_exported(L):
    for name in header_symbols():
        assert hasattr(L, name), name
    assert L.sr355_abi_version() == _lib.ABI_VERSION


def test_struct_sizes_match_header_layout():
    # natural alignment on x86-64: see sr355_conv_args / sr355_generator_cfg in include/sr355.h
    assert C.sizeof(GeneratorCfg) == 8 * 4
    assert C.sizeof(ConvArgs) % 8 == 0 and C.sizeof(ConvArgs) >= 26 * 4


def test_size_queries(L):
    assert L.sr355_unit_channels(_lib.DTYPE_BF16) == 8
    assert L.sr355_unit_channels(_lib.DTYPE_F32) == 4
    assert L.sr355_unit_channels(7) < 0
    # [cin_units][9][cout_pad][16 B]
    assert L.sr355_packed_weight_bytes(32, 64, _lib.DTYPE_BF16, 0) == 8 * 9 * 32 * 16
    assert L.sr355_packed_weight_bytes(64, 192, _lib.DTYPE_BF16, 0) == 24 * 9 * 64 * 16
    assert L.sr355_packed_weight_bytes(3, 64, _lib.DTYPE_F32, 0) == 16 * 9 * 32 * 16
    assert L.sr355_packed_weight_bytes(64, 3, _lib.DTYPE_BF16, 0) == 2 * 9 * 64 * 16
    assert L.sr355_packed_weight_bytes(0, 3, _lib.DTYPE_BF16, 0) == 0
    # split modes append the residue image; input split counts 2*cin channels; bf16 only
    assert L.sr355_packed_weight_bytes(64, 64, _lib.DTYPE_BF16, _lib.SPLIT_WEIGHT) == 2 * 8 * 9 * 64 * 16
    assert L.sr355_packed_weight_bytes(64, 3, _lib.DTYPE_BF16, _lib.SPLIT_INPUT) == 2 * 2 * 9 * 64 * 16
    assert L.sr355_packed_weight_bytes(64, 64, _lib.DTYPE_F32, _lib.SPLIT_WEIGHT) == 0
    cfg = GeneratorCfg(3, 3, 64, 23, 32, _lib.DTYPE_BF16, 1, 1)
    assert L.sr355_generator_num_convs(C.byref(cfg)) == 351
    assert L.sr355_generator_packed_bytes(C.byref(cfg)) > 16697987 * 2
    small = L.sr355_generator_workspace_bytes(C.byref(cfg), 1, 16, 16)
    big = L.sr355_generator_workspace_bytes(C.byref(cfg), 16, 128, 128)
    assert 0 < small < big
    # proportional to the pixel count, apart from the chain kernel's table of passes (a fixed ~0.1 MB)
    assert big == pytest.approx((small - 120000) * 16 * 64, rel=0.02)


def test_bad_arguments_are_rejected_without_touching_a_device(L):
    bad = GeneratorCfg(3, 3, 60, 23, 32, _lib.DTYPE_BF16, 1, 1)      # nf not a multiple of 16
    assert L.sr355_generator_num_convs(C.byref(bad)) < 0
    assert b"multiples of 16" in L.sr355_last_error()
    assert L.sr355_generator_packed_bytes(C.byref(bad)) == 0
    assert L.sr355_conv3x3_fwd(None, None) == -1
    a = ConvArgs()
    a.dtype, a.N, a.H, a.W, a.cout = _lib.DTYPE_BF16, 1, 8, 8, 32
    a.cin_units, a.in_units = 3, 8                                  # odd unit count
    assert L.sr355_conv3x3_fwd(C.byref(a), None) == -1
    assert b"cin_units" in L.sr355_last_error()
    a.cin_units = 8                                                 # null input pointer
    assert L.sr355_conv3x3_fwd(C.byref(a), None) == -1
    assert b"input buffer" in L.sr355_last_error()
    a.upsample, a.H = 1, 7                                          # odd output under x2 upsample
    assert L.sr355_conv3x3_fwd(C.byref(a), None) == -1
    cfg = GeneratorCfg(3, 3, 64, 1, 32, _lib.DTYPE_BF16, 1, 0)
    assert L.sr355_generator_forward(C.byref(cfg), None, 0, None, None, 1, 8, 8, None, 0, None) == -1


def test_integration_snippet_quotes_the_header_constants():
    """The ctypes stub in the integration notes is what a maintainer pastes: its ABI version, dtype code and config fields must
    be the header's."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    hdr = open(os.path.join(root, "include", "sr355.h")).read()
    abi = int(re.search(r"#define SR355_ABI_VERSION (\d+)", hdr).group(1))
    bf16 = int(re.search(r"#define SR355_DTYPE_BF16 (\d+)", hdr).group(1))
    f32 = int(re.search(r"#define SR355_DTYPE_F32 (\d+)", hdr).group(1))
    assert int(re.search(r"sr355_abi_version\(\) == (\d+)", doc).group(1)) == abi
    m = re.search(r"cfg = Cfg\(([\d, ]+)\)\s*# dtype (\d+) = SR355_DTYPE_BF16 \((\d+) = SR355_DTYPE_F32\)", doc)
    assert m, "the snippet's Cfg(...) line with its dtype comment"
    values = [int(v) for v in m.group(1).split(",")]
    fields = re.search(r'for n in \(([^)]*)\)\]', doc, re.S).group(1)
    names = re.findall(r'"(\w+)"', fields)
    struct = re.search(r"typedef struct sr355_generator_cfg \{(.*?)\} sr355_generator_cfg;", hdr, re.S).group(1)
    struct = re.sub(r"/\*.*?\*/", "", struct, flags=re.S)
    declared = [n.strip() for decl in struct.split(";") if decl.strip() for n in decl.replace("int", "").split(",")]
    assert names == declared and len(values) == len(names)
    assert values[names.index("dtype")] == bf16 == int(m.group(2)) and int(m.group(3)) == f32
