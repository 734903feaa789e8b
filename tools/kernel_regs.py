"""Register / spill / LDS table of the kernels of one built object (mm-diffusion_amd/lib/<name>.o), from the code object's metadata notes.
usage: python tools/kernel_regs.py mmd_gemm [name filter]"""
import os, re, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    obj = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mm-diffusion_amd", "lib", sys.argv[1] + ".o")
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    with tempfile.TemporaryDirectory() as d:
        os.symlink(os.path.abspath(obj), os.path.join(d, "k.o"))
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", "k.o"], cwd=d, check=True, capture_output=True)
        co = [f for f in os.listdir(d) if "gfx950" in f][0]
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], cwd=d, check=True, capture_output=True, text=True).stdout
        filt = subprocess.run(["c++filt"], input=notes, capture_output=True, text=True).stdout
    rows = []
    for blk in re.split(r"\n\s+- \.agpr_count", filt)[1:]:
        g = lambda k: re.search(r"\." + k + r":\s+(\S+)", blk)
        name = re.search(r"\.name:\s+(.*)", blk).group(1).strip().strip("'")
        if flt in name:
            rows.append((name[:110], g("vgpr_count").group(1), g("vgpr_spill_count").group(1), g("sgpr_count").group(1),
                         g("group_segment_fixed_size").group(1), g("private_segment_fixed_size").group(1)))
    for r in sorted(rows):
        print(f"{r[0]:110s} vgpr {r[1]:>3s} spill {r[2]:>3s} sgpr {r[3]:>3s} lds {r[4]:>6s} scratch {r[5]:>4s}")


if __name__ == "__main__":
    main()
