"""Registers, LDS and scratch of every kernel in the built library (from the code object's metadata notes).

usage: python tools/kernel_resources.py [path/to/libaprilgrid_amd.so]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def notes(lib):
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib, os.path.join(tmp, "unused")],
                       check=True, capture_output=True)
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True, capture_output=True)
        return subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "aprilgrid-rs_amd", "libaprilgrid_amd.so")
    text = notes(lib)
    for blk in text.split("  - .agpr_count:")[1:]:
        def g(key):
            m = re.search(r"\.%s:\s+(\S+)" % key, blk)
            return m.group(1) if m else "?"
        name = g("name")
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
        print("%-46s vgpr %4s sgpr %4s lds %6s scratch %5s max_wg %5s" % (name, g("vgpr_count"), g("sgpr_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size"), g("max_flat_workgroup_size")))


if __name__ == "__main__":
    main()
