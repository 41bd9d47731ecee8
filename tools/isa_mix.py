#!/usr/bin/env python3
"""Static instruction mix of one kernel of a built library (llvm-objdump -d of its gfx950 code object): the top opcodes and a summary
by class.  The cycle kernels are almost straight-line outside the QP loop, so the static mix is close to what one instance issues.
    python tools/isa_mix.py <kernel name filter> [libdwbc_amd/libdwbc_hip.so] [top N]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin/"
filt = sys.argv[1]
lib = sys.argv[2] if len(sys.argv) > 2 else "libdwbc_amd/libdwbc_hip.so"
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40


def klass(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if re.match(r"v_(fma|fmac|mul|add|sub|rcp|rsq|sqrt|div|max|min|trunc|floor|fract|ldexp|frexp|cvt|sin|cos|exp|log)_?.*f(64|32)", op) or "_f64" in op or "_f32" in op:
        return "VALU float (f64/f32 arithmetic, conversions, compares)"
    if op.startswith("v_readlane") or op.startswith("v_readfirstlane") or op.startswith("v_writelane") or "dpp" in op or op.startswith("v_permlane") or op.startswith("ds_bpermute") or op.startswith("ds_swizzle"):
        return "cross-lane (readlane / writelane / dpp)"
    if op.startswith("v_cndmask"):
        return "VALU select (v_cndmask)"
    if op.startswith("v_mov") or op.startswith("v_accvgpr"):
        return "VALU move"
    if op.startswith("v_"):
        return "VALU integer / logic / compare"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier") or op.startswith("s_sleep"):
        return "wait / nop"
    if op.startswith("s_cbranch") or op.startswith("s_branch") or op.startswith("s_setpc") or op.startswith("s_swappc") or op.startswith("s_endpgm"):
        return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer") or op.startswith("s_memtime") or op.startswith("s_memrealtime") or op.startswith("s_dcache"):
        return "scalar memory"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith("global_") or op.startswith("flat_") or op.startswith("buffer_") or op.startswith("scratch_"):
        return "vector memory"
    return "other"


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
        dis = subprocess.run([LLVM + "llvm-objdump", "-d", "--demangle", co], capture_output=True, text=True).stdout
        cur, ops = None, collections.Counter()
        done = False
        for line in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
            if m:
                if cur and ops:
                    done = True
                    break
                cur = m.group(1) if (filt in m.group(1) and "(" in m.group(1)) else None
                continue
            if cur:
                mm = re.match(r"^\s+([a-z_0-9]+)\s", line)
                if mm:
                    ops[mm.group(1)] += 1
        if cur and ops:
            done = True
        if done:
            tot = sum(ops.values())
            print(f"static instruction mix of {cur[:140]}: {tot} instructions")
            cl = collections.Counter()
            for op, n in ops.items():
                cl[klass(op)] += n
            for k, n in cl.most_common():
                print(f"  {n:7d}  {100.0 * n / tot:5.1f} %  {k}")
            print("top opcodes:")
            for op, n in ops.most_common(top):
                print(f"  {n:7d}  {op}")
            break
