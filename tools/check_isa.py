"""Invariants of the compiled K1 (k_blur_hessian) that the source can only ask for, checked on the code
object inside the built library (llvm-objdump of its gfx950 image):

K1 polls the frame's published minimum with a scalar `s_load_dword ... glc` issued by inline asm.  The
first poll is issued at wave start and awaited seven rows later; the compiler does not know that its
destination register is pending in between, so the register must not be read, written or copied
before the first `s_waitcnt lgkmcnt(0)` that follows -- otherwise the late result would land in a
register that holds something else by then.  Every later poll waits inside its own asm block.

usage: python tools/check_isa.py [path/to/libaprilgrid_amd.so]      exit status 1 on a violation"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def disassemble(lib):
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib, os.path.join(tmp, "unused")],
                       check=True, capture_output=True)
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True, capture_output=True)
        return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], check=True, capture_output=True, text=True).stdout


def check(lib):
    text = disassemble(lib).split("\n")
    starts = [i for i, l in enumerate(text) if re.match(r"^[0-9a-f]+ <\w+>:", l)]
    starts.append(len(text))
    problems, checked = [], 0
    for a, b in zip(starts[:-1], starts[1:]):
        name = re.match(r"^[0-9a-f]+ <(\w+)>:", text[a]).group(1)
        if "k_blur_hessian" not in name:
            continue
        body = text[a + 1:b]
        polls = [(i, re.search(r"s_load_dword (s\d+), .*\bglc\b", l)) for i, l in enumerate(body)]
        polls = [(i, m.group(1)) for i, m in polls if m]
        if not polls:
            problems.append("%s: no scalar glc poll found" % name)
            continue
        checked += 1
        regs = sorted(set(r for _, r in polls))
        if len(regs) != 1:
            problems.append("%s: polled value lives in several registers %s" % (name, regs))
        first, reg = polls[0]
        k = first + 1
        while k < len(body) and not re.search(r"s_waitcnt.*lgkmcnt\(0\)", body[k]):
            if re.search(r"\b%s\b" % reg, body[k]):
                problems.append("%s: %s touched while the first poll is pending: %s" % (name, reg, body[k].strip()))
            k += 1
        if k == len(body):
            problems.append("%s: no lgkmcnt(0) wait after the first poll" % name)
        for i, r in polls[1:]:  # later polls: awaited at once
            if not re.search(r"s_waitcnt.*lgkmcnt\(0\)", body[i + 1]):
                problems.append("%s: poll at +%d is not followed by its wait" % (name, i))
    if checked < 8:
        problems.append("only %d k_blur_hessian variants found in the code object" % checked)
    return checked, problems


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "aprilgrid-rs_amd", "libaprilgrid_amd.so")
    n, problems = check(lib)
    for p in problems:
        print("ISA CHECK:", p)
    print("%d k_blur_hessian variants checked, %d problems" % (n, len(problems)))
    sys.exit(1 if problems else 0)
