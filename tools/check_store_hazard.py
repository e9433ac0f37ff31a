#!/usr/bin/env python3
"""Static check of the compiled kernels for the wide-store data hazard of gfx950.

A vector-memory store of more than 64 bits keeps reading its data registers after it has issued: the ISA asks for two
wait states before a VALU instruction may overwrite them.  hipcc (ROCm 7.2) inserts them when the store's scalar offset
is an immediate, but not when it is an SGPR (its hazard recogniser exempts that form), and never inside an inline-asm
statement.  On the hardware the exemption does not hold: the discriminator's 64 -> 128 conv came out with a few
elements of stale register contents about once in forty launches on a cold memory system (the first dword of the
store, overwritten by the very next instruction).

This tool disassembles every code object of the library (or compiles the sources to assembly with --compile) and lists
every dwordx3 / dwordx4 store whose data registers are written by a VALU instruction less than two wait states later.
Exit status 1 if there is one.

python3 tools/check_store_hazard.py [libsr355.so ...]      python3 tools/check_store_hazard.py --compile"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "super_resolution_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"
NEED = 2                                  # wait states between the store and a write of its data registers

_STORE = re.compile(r"\b(buffer|global|flat|scratch)_store_(dwordx[34]|format_xyzw?|b96|b128)\s+(.*)")
_VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
_LABEL = re.compile(r"^[A-Za-z_.$][\w.$]*:")


def _regs(tok):
    m = _VREG.search(tok)
    if not m:
        return None
    if m.group(1) is not None:
        return int(m.group(1)), int(m.group(1))
    return int(m.group(2)), int(m.group(3))


def _data_regs(mnemonic_class, operands):
    """The data operand of a wide store: the first vector operand of a buffer store, the second of a global / flat one."""
    toks = [t.strip() for t in operands.split(",")]
    vec = [t for t in toks if _VREG.fullmatch(t.split()[0]) or t.startswith("v[")]
    if not vec:
        return None
    if mnemonic_class == "buffer":
        return _regs(vec[0])
    wide = [t for t in vec if t.startswith("v[")]
    # global_store_dwordx4 vaddr, vdata, saddr: the data operand is the 3 / 4 register tuple
    for t in wide:
        lo, hi = _regs(t)
        if hi - lo + 1 >= 3:
            return lo, hi
    return None


def instructions(text):
    """(kernel, line number, instruction text) for every instruction; labels reset nothing (fall-through is the
    common case and a branch target only adds distance)."""
    kernel = None
    for no, raw in enumerate(text.split("\n"), 1):
        line = raw.split(";")[0].split("//")[0].strip()
        if not line:
            continue
        m = re.match(r"^([0-9a-f]+ )?<?(_Z\w+|[A-Za-z_]\w*)>?:$", line)
        if m:
            if m.group(2).startswith("_Z") or not line.startswith("."):
                kernel = m.group(2)
            continue
        if _LABEL.match(line) or line.startswith("."):
            continue
        # objdump lines: "\ts_nop 1    // 000000001234: BF800001"
        yield kernel, no, line


def scan(text, name):
    ins = list(instructions(text))
    bad = []
    n_wide = 0
    for i, (kern, no, line) in enumerate(ins):
        m = _STORE.search(line)
        if not m:
            continue
        data = _data_regs(m.group(1), m.group(3))
        if data is None:
            continue
        n_wide += 1
        lo, hi = data
        ws = 0
        for _k, no2, nxt in ins[i + 1:i + 1 + NEED + 6]:
            if ws >= NEED:
                break
            op = nxt.split()[0]
            if op == "s_nop":
                ws += int(nxt.split()[1], 0) + 1
                continue
            if op.startswith("v_") and not op.startswith("v_cmp") and not op.startswith("v_readlane") \
                    and not op.startswith("v_readfirstlane"):
                dst = _regs(nxt.split(None, 1)[1].split(",")[0]) if len(nxt.split(None, 1)) > 1 else None
                if dst and not (dst[1] < lo or dst[0] > hi):
                    bad.append((name, kern, no, line, ws, nxt))
                    break
            ws += 1
    return n_wide, bad


def disassemble(lib):
    """Assembly text of every gfx950 code object bundled in a host shared object."""
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--list", "--type=o", f"--input={lib}"],
                           capture_output=True, text=True)
        # a fat binary inside an ELF: take the .hip_fatbin section apart instead
        fat = os.path.join(tmp, "fatbin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat],
                       check=True)
        blob = open(fat, "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
        for k, s in enumerate(starts):
            part = os.path.join(tmp, f"bundle{k}")
            open(part, "wb").write(blob[s:starts[k + 1] if k + 1 < len(starts) else len(blob)])
            lst = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--list", "--type=o", f"--input={part}"],
                                 capture_output=True, text=True).stdout.split()
            for target in lst:
                if "gfx950" not in target:
                    continue
                co = os.path.join(tmp, f"co{k}.elf")
                subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={part}",
                                f"--targets={target}", f"--output={co}"], check=True)
                out.append(subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co],
                                          capture_output=True, text=True, check=True).stdout)
    return out


def compile_sources():
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    tmp = tempfile.mkdtemp()
    procs = []
    for f in srcs:
        o = os.path.join(tmp, f[:-4] + ".s")
        procs.append((f, o, subprocess.Popen(
            ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + CSRC,
             "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", os.path.join(CSRC, f), "-o", o],
            stderr=subprocess.DEVNULL)))
    texts = []
    for f, o, p in procs:
        if p.wait():
            raise SystemExit(f"compiling {f} failed")
        texts.append((f, open(o).read()))
    return texts


def check_library(lib):
    total, bad = 0, []
    for k, text in enumerate(disassemble(lib)):
        n, b = scan(text, f"{os.path.basename(lib)}#{k}")
        total += n
        bad += b
    return total, bad


def main():
    args = sys.argv[1:]
    total, bad = 0, []
    if args and args[0] == "--compile":
        for f, text in compile_sources():
            n, b = scan(text, f)
            total += n
            bad += b
    else:
        for lib in args or [os.path.join(ROOT, "super_resolution_amd", "libsr355.so")]:
            n, b = check_library(lib)
            total += n
            bad += b
    for name, kern, no, line, ws, nxt in bad:
        print(f"{name}: {kern}: line {no}: {line}   overwritten after {ws} wait state(s) by: {nxt}")
    print(f"{total} wide stores, {len(bad)} with their data registers overwritten inside {NEED} wait states")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
