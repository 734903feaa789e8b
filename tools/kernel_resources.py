#!/usr/bin/env python3
"""Static per-kernel resource table (no GPU needed): compiles every csrc/*.hip for gfx950 with
-Rpass-analysis=kernel-resource-usage and tabulates VGPR / AGPR / scratch / LDS / occupancy per kernel.

    python tools/kernel_resources.py [> profiles/rNN_x_kernel_resources.txt]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-diffusion_amd"))
import build as mmd_build  # noqa: E402

KEYS = ["VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "VGPRs Spill", "LDS Size [bytes/block]"]


def demangle(names):
    # GNU c++filt does not know the bf16 mangling (DF16b): substitute a placeholder type for display
    try:
        out = subprocess.run(["c++filt"], input="\n".join(n.replace("DF16b", "8bfloat16") for n in names), capture_output=True, text=True,
                             check=True).stdout
        return [re.sub(r"\(.*", "", l.replace("void ", "")) for l in out.splitlines()]
    except Exception:
        return names


def main():
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        procs = []
        for s in mmd_build.SOURCES:
            cmd = ["/opt/rocm/bin/hipcc", *mmd_build.FLAGS, "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(mmd_build.SRC, s),
                   "-o", os.path.join(tmp, s + ".o")]
            procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        for s, p in procs:
            out, _ = p.communicate()
            cur = None
            for line in out.splitlines():
                m = re.search(r"remark:\s+Function Name: (\S+)", line)
                if m:
                    cur = {"file": s, "name": m.group(1)}
                    rows.append(cur)
                    continue
                m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\S+) \[-Rpass", line)
                if m and cur is not None:
                    cur[m.group(1).strip()] = m.group(2)
    for r, d in zip(rows, demangle([r["name"] for r in rows])):
        r["short"] = d
    print("# static resource usage per kernel, gfx950, flags: " + " ".join(mmd_build.FLAGS))
    print(f"{'file':24s} {'vgpr':>5s} {'agpr':>5s} {'scratch':>7s} {'spill':>5s} {'lds_B':>7s} {'occ':>4s}  kernel")
    for r in sorted(rows, key=lambda r: (r["file"], r["short"])):
        print(f"{r['file']:24s} {r.get(KEYS[0], '?'):>5s} {r.get(KEYS[1], '?'):>5s} {r.get(KEYS[2], '?'):>7s} {r.get(KEYS[4], '?'):>5s} "
              f"{r.get(KEYS[5], '?'):>7s} {r.get(KEYS[3], '?'):>4s}  {r['short']}")


if __name__ == "__main__":
    main()
