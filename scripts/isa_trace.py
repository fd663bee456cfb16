"""Walks ONE execution path of a kernel listing (a .s file stripped of directives) from a start label until it returns
to it, following the branch decisions given in a JSON file {"<label>#<n-th branch in that block>": true|false}
(default: not taken; unconditional branches are always followed), and prints per class the number of instructions
executed -- a dynamic count for the chosen path.  Used for profiles/r3_search_isa_breakdown.md (the usual C2 expansion).
    python scripts/isa_trace.py listing.s .LBB28_33 decisions.json [-v]"""
import collections
import json
import re
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from isa_breakdown import classify


def main():
    lines = [l.rstrip("\n") for l in open(sys.argv[1], errors="replace")]
    start = sys.argv[2]
    dec = json.load(open(sys.argv[3])) if len(sys.argv) > 3 and not sys.argv[3].startswith("-") else {}
    verbose = "-v" in sys.argv
    labels = {}
    local = collections.defaultdict(list)          # inline-asm local labels "1:" -> line numbers
    for i, l in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
        m = re.match(r"^\s*(\d+):\s*(;.*)?$", l)
        if m:
            local[m.group(1)].append(i)

    def resolve(tgt, at):
        m = re.match(r"^(\d+)([bf])$", tgt)
        if not m:
            return labels[tgt], tgt
        cands = local[m.group(1)]
        if m.group(2) == "b":
            return max(c for c in cands if c <= at), None
        return min(c for c in cands if c > at), None
    pc = labels[start]
    cur = start
    nbr = 0
    counts = collections.Counter()
    seen_ann = collections.Counter()
    visited = []
    steps = 0
    while steps < 20000:
        steps += 1
        pc += 1
        l = lines[pc].strip()
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            cur = m.group(1)
            nbr = 0
            if cur == start:
                break
            visited.append(cur)
            continue
        if not l or l.startswith((";", ".", "//")) or re.match(r"^\d+:\s*(;.*)?$", l):
            continue
        parts = l.split(None, 1)
        op = parts[0]
        if not re.match(r"^[a-z]", op):
            continue
        operands = parts[1].split(";")[0].strip() if len(parts) > 1 else ""
        counts[classify(op, operands)] += 1
        if verbose:
            print(f"{cur:14s} {l}")
        if op == "s_branch":
            tgt = operands.strip()
            if tgt == start:
                break
            cur = tgt; pc = labels[tgt]; nbr = 0; visited.append(cur)
        elif op.startswith("s_cbranch"):
            key = f"{cur}#{nbr}"
            nbr += 1
            ann = l.split(";")[-1].strip() if ";" in l else ""
            seen_ann[ann] += 1
            akey = f"{ann}#{seen_ann[ann]}"          # k-th branch attributed to that source line along this path
            taken = dec.get(key, dec.get(akey, dec.get(ann, False)))
            if key not in dec and (akey in dec or ann in dec):
                key = akey if akey in dec else ann
            if verbose or key not in dec:
                ctx = [x.strip() for x in lines[max(0, pc - 3):pc] if x.strip() and not x.strip().endswith(":")]
                for x in ctx:
                    print("          " + x)
                print(f"    [{key}] [{akey}] {l}   -> {'TAKEN' if taken else 'not taken'}{'' if key in dec else '   (default)'}")
            if taken:
                tgt = operands.strip()
                if tgt == start:
                    break
                pc, name = resolve(tgt, pc)
                if name:
                    cur = name; nbr = 0; visited.append(cur)
        elif op == "s_endpgm":
            break
    print("blocks:", " ".join(visited))
    tot = sum(counts.values())
    for k, v in sorted(counts.items(), key=lambda kv: -kv[1]):
        print(f"{k:22s} {v}")
    salu = sum(v for k, v in counts.items() if k.startswith("SALU"))
    print(f"total {tot}   SALU {salu}   VALU {counts['VALU']}  lane<->scalar {counts['lane<->scalar']}")


if __name__ == "__main__":
    main()
