"""Registers / spills per kernel from a `-Rpass-analysis=kernel-resource-usage` log (hipcc stderr):
    python scripts/resource_usage.py build.log [filter]"""
import re
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cur = {}
for line in txt.splitlines():
    m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": v}
    cur[k] = v
    if k.startswith("LDS Size") and flt in cur["name"]:
        print(f"{cur['name'][:70]:70s} vgpr {cur.get('VGPRs'):>4s} occ {cur.get('Occupancy [waves/SIMD]'):>2s} sgpr-spill {cur.get('SGPRs Spill'):>3s} vgpr-spill {cur.get('VGPRs Spill'):>3s} scratch {cur.get('ScratchSize [bytes/lane]')}")
