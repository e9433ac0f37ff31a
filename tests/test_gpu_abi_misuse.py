"""GPU: the C ABI under misuse.  Every entry point of include/sr355.h is called through ctypes with (a) all-zero arguments,
(b) plausible sizes but NULL buffers and (c) the same with a valid arithmetic code in the argument blocks, so that the checks
behind the first one are reached; each must come back with an error code (or a zero size) and a message, never a crash or a
launch.  The sweep runs in a child process, so a crash would show as its exit status instead of taking the test
run down; afterwards the child runs one real call to show the library still works."""
import ctypes as C
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SKIP = {"sr355_abi_version", "sr355_last_error", "sr355_chain_timing", "sr355_chain_last_ms", "sr355_chain_status",
        "sr355_chain_reenable", "sr355_unit_channels"}


def _fill(ctype, mode):
    """A value of ``ctype``: zeros (mode 0) or small sizes with NULL pointers (mode 1)."""
    if ctype in (C.c_void_p, C.c_char_p) or (isinstance(ctype, type) and issubclass(ctype, C._Pointer)):
        if mode and isinstance(ctype, type) and issubclass(ctype, C._Pointer) and issubclass(ctype._type_, C.Structure):
            s = ctype._type_()
            for name, ft in s._fields_:
                if isinstance(ft, type) and issubclass(ft, C.Array):
                    continue                                   # pointer / offset tables stay zero
                if ft in (C.c_void_p,) or (isinstance(ft, type) and issubclass(ft, C._Pointer)):
                    continue
                setattr(s, name, 1.0 if ft in (C.c_float, C.c_double) else (1 << 20 if ft is C.c_size_t else 8 * mode))
                if mode == 2 and name in ("dtype", "hr_split", "trunk_fp32", "upsample", "accumulate", "stride"):
                    setattr(s, name, 0 if name != "stride" else 1)
            return C.pointer(s)
        return None
    if ctype in (C.c_float, C.c_double):
        return 1.0 if mode else 0.0
    if ctype is C.c_size_t:
        return (1 << 20) if mode else 0
    return 8 * mode


def sweep():
    sys.path.insert(0, ROOT)
    from super_resolution_amd import _lib
    L = _lib.lib()
    calls = 0
    for name in _lib.EXPORTS:
        if name in SKIP:
            continue
        fn = getattr(L, name)
        for mode in (0, 1, 2):
            args = [_fill(t, mode) for t in (fn.argtypes or [])]
            rc = fn(*args)
            calls += 1
            if fn.restype is C.c_size_t:
                continue                                       # a size query answers 0 for what it cannot size
            if fn.restype is None:
                continue
            if rc == 0:
                print(f"{name} mode {mode}: accepted {args}")
                return 2
            if not L.sr355_last_error():
                print(f"{name} mode {mode}: error {rc} without a message")
                return 3
    # the library still works: one real conversion round trip
    import torch
    from super_resolution_amd import ops
    x = torch.rand(1, 8, 5, 6, device="cuda:0")
    back = ops.from_blocked(ops.to_blocked(x, 1, _lib.DTYPE_BF16), 0, 8)
    if (back - x).abs().max().item() > 1e-2:
        print("round trip after the sweep is wrong")
        return 4
    print(f"swept {calls} calls")
    return 0


@pytest.mark.gpu
def test_every_entry_point_refuses_null_and_zero_arguments():
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.abspath(__file__)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    assert "swept" in r.stdout


if __name__ == "__main__":
    sys.exit(sweep())
