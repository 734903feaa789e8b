#!/usr/bin/env python3
"""Build check for the pipelined row-strip GEMM (mmd_gemm.hip, conv1x1_strip_res_kernel, K >= 384): its residual pieces live in v[248:255],
registers the compiler must never touch (amdgpu_num_vgpr(248)); they are written by `global_load_dwordx4 v[248:251] / v[252:255]` and read
only by the unpack instructions that follow an `s_waitcnt vmcnt` in the same asm statement.  This script disassembles the built object and
fails if any OTHER instruction of those kernels names one of the eight registers, or if an unpack is not preceded by the wait.

    python tools/strip_asm_check.py [mm-diffusion_amd/lib/mmd_gemm.o]      (run by tests/test_host_cpu.py when the object exists)"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get("LLVM_BIN", "/opt/rocm/lib/llvm/bin")


def disassemble(obj):
    with tempfile.TemporaryDirectory() as d:
        os.symlink(os.path.abspath(obj), os.path.join(d, "k.o"))
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", "k.o"], cwd=d, check=True, capture_output=True)
        co = [f for f in os.listdir(d) if "gfx950" in f][0]
        return subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--mcpu=gfx950", co], cwd=d, check=True, capture_output=True, text=True).stdout


def regs_of(args):
    out = set()
    for m in re.finditer(r"v\[(\d+):(\d+)\]|\bv(\d+)\b", args):
        if m.group(1) is not None:
            out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def check(asm):
    """-> (kernels checked, list of violations)"""
    kernels, bad = 0, []
    cur, ins = None, []

    def flush():
        nonlocal kernels
        if not cur or "conv1x1_strip_res_kernel" not in cur:
            return
        m = re.search(r"conv1x1_strip_res_kernelILi(\d+)E", cur)
        if not m or int(m.group(1)) < 6:          # K = 384 / 512: the pipelined instances with a residual
            return
        kernels += 1
        loads = 0
        for i, (addr, op, args) in enumerate(ins):
            r = regs_of(args) & set(range(248, 256))
            if not r:
                continue
            if op == "global_load_dwordx4" and re.match(r"v\[(248:251|252:255)\],", args):
                loads += 1
                continue
            if op in ("v_lshlrev_b32_e32", "v_and_b32_e32") and len(r) == 1:
                # an unpack of the asm statement: the statement starts with the wait, at most 8 instructions back
                back = [ins[j][1] + " " + ins[j][2] for j in range(max(0, i - 8), i)]
                if any(b.startswith("s_waitcnt vmcnt") for b in back):
                    continue
                bad.append(f"{cur[:60]}: {addr:x} {op} {args}: unpack without the wait in front")
                continue
            bad.append(f"{cur[:60]}: {addr:x} {op} {args}: touches v[248:255]")
        if loads == 0:
            bad.append(f"{cur[:60]}: no residual request found - the check looked at the wrong code")

    for line in asm.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            flush()
            cur, ins = m.group(1), []
            continue
        m = re.match(r"\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):", line)
        if m:
            ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
    flush()
    return kernels, bad


def main():
    obj = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mm-diffusion_amd", "lib", "mmd_gemm.o")
    n, bad = check(disassemble(obj))
    for b in bad:
        print(b)
    print(f"{n} kernels checked, {len(bad)} violations")
    return 1 if bad or n == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
