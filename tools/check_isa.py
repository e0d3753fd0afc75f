#!/usr/bin/env python
"""Build gate: no kernel of libst2_hip.so may contain a packed-f32 VALU op whose `op_sel` takes the HIGH dword of a source
for the LOW result lane.

Why (round 6, tools/simd_hazard_repro.hip, profiles/r06*_hazard.log): on gfx950 (MI355X) `v_pk_fma_f32 / v_pk_mul_f32 /
v_pk_add_f32 ... op_sel:[0,1,..]` returns a wrong LOW result half in lanes 48-63 of the wave while another wave of the same
CU issues MFMAs in certain cadences (any v_mfma_f32_16x16x32_f16 stream; v_mfma_f32_32x32x16_f16 issued in isolated groups,
which is what a one-accumulator conv tile does).  Plain encodings, `op_sel_hi` broadcasts, op_sel on src0 / src2, scalar
v_fma_f32, LDS and global loads are not affected (0 of ~10^9).  hipcc emits the bad encoding when it packs two fp32
operations that share ONE operand sitting in the odd register of a 64-bit pair (the SLP vectorizer on {a*h, b*h} with h an
odd element of a ds_read_b128) -- rounds 1-5 had it in exactly two kernels, both BiLSTM recurrences, which is why only they
were ever "irreproducible next to narrow-tile convs".  Conservative rule: ANY op_sel bit on a packed-f32 op fails the gate.

Usage: python tools/check_isa.py [path/to/libst2_hip.so]   (exit status 1 and a listing when violated)
Pure host work: llvm-objdump from the ROCm toolchain, no GPU."""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP_CANDIDATES = ["/opt/rocm/lib/llvm/bin/llvm-objdump", "/opt/rocm/llvm/bin/llvm-objdump", "llvm-objdump"]
PACKED_F32 = re.compile(r"\b(v_pk_(?:fma|mul|add)_f32)\b")
OP_SEL = re.compile(r"\bop_sel:\[([01,]+)\]")


def _objdump():
    for c in OBJDUMP_CANDIDATES:
        p = c if os.path.isabs(c) else shutil.which(c)
        if p and os.path.exists(p):
            return p
    return None


def disassemble(lib_path):
    """{kernel symbol: [instruction lines]} over every gfx950 code object bundled in `lib_path`; None without llvm-objdump."""
    od = _objdump()
    if od is None:
        return None
    kernels = {}
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, os.path.basename(lib_path))
        shutil.copy(lib_path, local)  # --offloading extracts the bundles next to its input
        subprocess.run([od, "--offloading", local], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for co in sorted(glob.glob(local + ".*amdgcn*")):
            text = subprocess.run([od, "-d", co], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
            cur = None
            for line in text.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
                if m:
                    cur = m.group(1)
                    kernels.setdefault(cur, [])
                elif cur is not None and line.strip():
                    kernels[cur].append(line.split("//")[0].strip())
    return kernels


def violations(kernels):
    """[(kernel, instruction)] for packed-f32 ops with any op_sel bit set."""
    out = []
    for k, lines in kernels.items():
        for ins in lines:
            if PACKED_F32.search(ins):
                m = OP_SEL.search(ins)
                if m and "1" in m.group(1):
                    out.append((k, ins))
    return out


def check(lib_path):
    """(ok, report): ok is None when the disassembler is missing (nothing checked)."""
    kernels = disassemble(lib_path)
    if kernels is None:
        return None, "llvm-objdump not found: ISA gate skipped"
    bad = violations(kernels)
    n_pk = sum(1 for lines in kernels.values() for ins in lines if PACKED_F32.search(ins))
    if not bad:
        return True, "ISA gate: %d kernels, %d packed-f32 ops, none with op_sel" % (len(kernels), n_pk)
    by_kernel = {}
    for k, ins in bad:
        by_kernel.setdefault(k, []).append(ins)
    lines = ["ISA gate FAILED: %d packed-f32 op(s) with op_sel (wrong low half in lanes 48-63 next to MFMA waves on gfx950):" % len(bad)]
    for k, v in sorted(by_kernel.items(), key=lambda kv: -len(kv[1])):
        lines.append("  %4d  %s   e.g. %s" % (len(v), k[:110], v[0]))
    lines.append("fix: keep the shared operand in the LOW register of its pair (pack along K, see st2_lstm_coop.hip), or build the "
                 "translation unit with -fno-slp-vectorize")
    return False, "\n".join(lines)


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "styletts2_amd", "libst2_hip.so")
    ok, report = check(path)
    print(report)
    sys.exit(0 if ok is not False else 1)
