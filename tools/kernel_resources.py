#!/usr/bin/env python3
"""Per-kernel resource table of a .hip file (hipcc -Rpass-analysis=kernel-resource-usage): VGPRs, spills, occupancy, LDS.
  python tools/kernel_resources.py hypelcnn_amd/csrc/seg_gemm.hip [extra hipcc flags]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT}/include", "-c", src, "-o",
       "/tmp/_kr.o", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:]
out = subprocess.run(cmd, stderr=subprocess.PIPE, text=True).stderr
cur, rows = None, []
for line in out.splitlines():
    m = re.search(r"remark: +(Function Name|VGPRs|AGPRs|TotalSGPRs|SGPRs Spill|VGPRs Spill|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|ScratchSize \[bytes/lane\]): (\S+)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], stdout=subprocess.PIPE, text=True).stdout.strip()}
        rows.append(cur)
    elif cur is not None:
        cur[k.split(" [")[0]] = v
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::|\(.*$", "", r["name"])
    print(f"{name:70s} vgpr {r.get('VGPRs'):>4s} agpr {r.get('AGPRs', '0'):>3s} sgpr {r.get('TotalSGPRs'):>4s} "
          f"spill s/v {r.get('SGPRs Spill'):>3s}/{r.get('VGPRs Spill'):>3s} scratch {r.get('ScratchSize', '0'):>4s} "
          f"occ {r.get('Occupancy'):>2s} lds {r.get('LDS Size'):>6s}")
