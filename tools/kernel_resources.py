#!/usr/bin/env python3
"""Register / LDS / spill figures of the kernels in a built library, from the metadata notes of its gfx950 code objects.
    python tools/kernel_resources.py [libdwbc_amd/libdwbc_hip.so] [name filter ...]"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin/"
lib = sys.argv[1] if len(sys.argv) > 1 else "libdwbc_amd/libdwbc_hip.so"
filt = sys.argv[2:] or ["dwbc_cycle_kernel"]
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
        r = subprocess.run([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + part, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co],
                           capture_output=True)
        if r.returncode or not os.path.exists(co) or os.path.getsize(co) == 0:
            continue
        notes = subprocess.run([LLVM + "llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
        for blk in notes.split("- .agpr_count")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk)
            if not name:
                continue
            dem = subprocess.run(["c++filt", name.group(1)], capture_output=True, text=True).stdout.strip()
            if not all(f in dem for f in filt):
                continue
            g = lambda k: (re.search(r"\." + k + r":\s+(\d+)", blk) or [None, "?"])[1]
            agpr = re.match(r":\s+(\d+)", blk)
            print(f"{dem[:118]:118s} vgpr {g('vgpr_count'):>4s} agpr {agpr.group(1) if agpr else '?':>3s} sgpr {g('sgpr_count'):>3s} vspill {g('vgpr_spill_count'):>4s} "
                  f"sspill {g('sgpr_spill_count'):>4s} scratch {g('private_segment_fixed_size'):>5s}")
