#!/usr/bin/env python
"""Which kernel instantiations of the library use scratch memory (register spills / private arrays): compiles every
csrc/*.hip for gfx950 with --save-temps and lists the kernels whose .private_segment_fixed_size is non-zero.
    python tools/spills.py > profiles/r03_scratch_by_kernel.txt"""
import glob
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = []
with tempfile.TemporaryDirectory() as tmp:
    for src in sorted(glob.glob(os.path.join(ROOT, "mdgrad_amd", "csrc", "*.hip"))):
        base = os.path.basename(src)[:-4]
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", src, "-o", base + ".o",
                        "--save-temps"], cwd=tmp, capture_output=True)
        asm = os.path.join(tmp, base + "-hip-amdgcn-amd-amdhsa-gfx950.s")
        if not os.path.exists(asm):
            continue
        total = 0
        for b in open(asm).read().split("  - .agpr_count:")[1:]:
            n = re.search(r"\.name:\s+(\S+)", b)
            sz = re.search(r"\.private_segment_fixed_size:\s+(\d+)", b)
            vg = re.search(r"\.vgpr_count:\s+(\d+)", b)
            total += 1
            if n and sz and int(sz.group(1)) > 0:
                d = subprocess.run(["c++filt", n.group(1)], capture_output=True, text=True).stdout.strip()
                d = d.replace("(anonymous namespace)::", "").replace("void ", "")
                rows.append((base + ".hip", int(sz.group(1)), int(vg.group(1)), d.split("(")[0]))
        rows.append((base + ".hip", -1, total, ""))
print("# kernels with scratch (bytes per lane), gfx950, hipcc -O3; files without a row below have none")
print("%-20s %8s %6s  %s" % ("file", "scratch", "vgpr", "kernel"))
for f, sz, vg, name in rows:
    if sz >= 0:
        print("%-20s %8d %6d  %s" % (f, sz, vg, name))
print("# kernel instantiations per file: " + ", ".join("%s %d" % (f, vg) for f, sz, vg, _ in rows if sz < 0))
