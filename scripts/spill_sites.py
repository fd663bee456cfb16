"""Where a kernel's spill code sits: scratch loads/stores and SGPR<->VGPR lane moves per source line of one kernel
   hipcc ... --offload-device-only -S -gline-tables-only -o k.s ;  python scripts/spill_sites.py k.s '<mangled kernel name>'"""
import re
import sys
from collections import Counter

txt = open(sys.argv[1]).read()
name = sys.argv[2]
m = re.search(r"^%s:(.*?)^\s*\.end_amdhsa_kernel" % re.escape(name), txt, re.S | re.M)
body = m.group(1)
files = dict(re.findall(r'\.file\s+(\d+)\s+"[^"]*?([^"/]+)"\s*$', txt, re.M))
cur = None
c = Counter()
n = 0
for ln in body.splitlines():
    mm = re.search(r"\.loc\s+(\d+)\s+(\d+)", ln)
    if mm:
        cur = (files.get(mm.group(1), mm.group(1)), int(mm.group(2)))
    t = ln.strip()
    if not t or t.startswith((".", ";", "//")) or t.endswith(":"):
        continue
    n += 1
    if re.match(r"scratch_(load|store)", t):
        c[("scratch", cur)] += 1
    elif re.match(r"v_(readlane|writelane)_b32", t):
        c[("lane", cur)] += 1
print("instructions", n, "scratch", sum(v for k, v in c.items() if k[0] == "scratch"), "lane moves", sum(v for k, v in c.items() if k[0] == "lane"))
if len(sys.argv) > 3:
    for k, v in sorted(c.items(), key=lambda x: (x[0][0], str(x[0][1]))):
        print(k, v)
