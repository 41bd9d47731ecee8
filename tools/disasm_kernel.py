#!/usr/bin/env python3
"""Disassembly of one kernel of a built library (llvm-objdump -d of its gfx950 code object) to a text file.
    python tools/disasm_kernel.py <kernel name filter> <library.so> <out.s>"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin/"
filt, lib, out = sys.argv[1], sys.argv[2], sys.argv[3]
with tempfile.TemporaryDirectory() as tmp:
    fat = os.path.join(tmp, "fat.bin")
    subprocess.check_call([LLVM + "llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
    blob = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
    for bi, st in enumerate(starts):
        part = os.path.join(tmp, f"b{bi}.bin")
        open(part, "wb").write(blob[st:starts[bi + 1] if bi + 1 < len(starts) else len(blob)])
        co = os.path.join(tmp, f"co{bi}.o")
        r = subprocess.run([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + part, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], capture_output=True)
        if r.returncode or not os.path.exists(co) or os.path.getsize(co) == 0:
            continue
        dis = subprocess.run([LLVM + "llvm-objdump", "-d", "--demangle", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
        for blk in re.split(r"\n(?=[0-9a-f]+ <)", dis):
            head = blk.split("\n", 1)[0]
            if filt in head:
                open(out, "w").write(blk)
                print(head, len(blk.splitlines()), "lines ->", out)
                sys.exit(0)
print("kernel not found")
sys.exit(1)
