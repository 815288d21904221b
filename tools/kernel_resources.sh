#!/bin/bash
# Per-kernel register / LDS / occupancy table of the engine (compile only; no GPU needed).
cd "$(dirname "$0")/../plant-3d-vision_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
  -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize -I../../include -c --cuda-device-only \
  -Rpass-analysis=kernel-resource-usage spacecarve.hip -o /dev/null 2>&1 |
python3 -c '
import re, sys, subprocess
cur = None
rows = []
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0]
        cur = {"name": name}
        rows.append(cur)
        continue
    for key in ("TotalSGPRs", "VGPRs", "AGPRs", "ScratchSize \[bytes/lane\]", "Occupancy \[waves/SIMD\]", "LDS Size \[bytes/block\]"):
        m = re.search(key + r": (\d+)", line)
        if m and cur is not None:
            cur[key.split(" ")[0].replace("\\\\", "")] = int(m.group(1))
print("%-44s %5s %5s %8s %5s %6s" % ("kernel", "SGPR", "VGPR", "scratch", "occ", "LDS"))
for r in rows:
    print("%-44s %5s %5s %8s %5s %6s" % (r["name"][:44], r.get("TotalSGPRs"), r.get("VGPRs"), r.get("ScratchSize"), r.get("Occupancy"), r.get("LDS")))
'
