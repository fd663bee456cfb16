"""Scratch reloads / spills inside the expansion loop (loop depth >= 2) of the search kernels -- each one is a VMEM
operation in the middle of the loads whose round trips the loop overlaps, so the count has to be zero on the usual path.
    hipcc ... --offload-device-only -S -gline-tables-only -o k.s ;  python scripts/loop_reloads.py k.s"""
import re
import sys

txt = open(sys.argv[1]).read()
for kn in re.findall(r"^(_ZN3cph13search_kernelILi\dELi\d+E(?:Lb\dE)?EEvNS_10SearchArgsE):", txt, re.M):
    m = re.search(r"^%s:(.*?)^\s*\.end_amdhsa_kernel" % re.escape(kn), txt, re.S | re.M)
    lines = m.group(1).splitlines()
    cur, depth, out = None, 0, []
    for i, ln in enumerate(lines):
        mm = re.search(r"\.loc\s+(\d+)\s+(\d+)", ln)
        if mm:
            cur = int(mm.group(2))
        lb = re.match(r"^\.LBB\S+:\s*;\s*(.*)", ln)
        if re.match(r"^\.LBB\S+:", ln):
            dm = re.search(r"Depth=(\d+)", ln)
            depth = int(dm.group(1)) if dm else 0
            if not dm and i + 1 < len(lines):
                dm = re.search(r"Depth=(\d+)", lines[i + 1])
                depth = int(dm.group(1)) if dm else 0
        if re.match(r"^\s*scratch_(load|store)", ln) and depth >= 2:
            out.append((cur, depth, ln.strip().split(";")[0].strip()))
    print(kn[17:30], "in-loop scratch ops:", len(out))
    for o in out:
        print("    line", o[0], "depth", o[1], o[2])
