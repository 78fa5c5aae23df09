#!/usr/bin/env python3
"""per-kernel register / scratch / LDS / occupancy figures of the product's HIP kernels (hipcc -Rpass-analysis), one line per kernel"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "ftk_amd", "csrc", "tile_kernels.hip")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-c", src,
       "-o", "/tmp/ftkx_res.o", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], {}
for line in out.splitlines():
    m = re.search(r"remark: +(\w[\w \[\]/]*): +(\S+)", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        if cur:
            rows.append(cur)
        cur = {"name": v}
    else:
        cur[k] = v
if cur:
    rows.append(cur)
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name).replace("void ", "")
    g = lambda k: r.get(k, "?")  # noqa: E731
    print(f"{name:58s} VGPR {g('VGPRs'):>4} AGPR {g('AGPRs'):>3} SGPR {g('TotalSGPRs'):>4} scratch {g('ScratchSize [bytes/lane]'):>5} "
          f"occ {g('Occupancy [waves/SIMD]'):>2} LDS {g('LDS Size [bytes/block]'):>6}")
