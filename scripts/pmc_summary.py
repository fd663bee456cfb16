"""Summarises rocprofv3 --pmc output directories (counter_collection.csv) per kernel:
    python scripts/pmc_summary.py gpurun_out/pmcA gpurun_out/pmcB --kernel search_kernel --per 2629998
prints mean counter value per dispatch and per unit (--per = e.g. expansions per dispatch)."""
import argparse
import csv
import glob
import os
from collections import defaultdict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("--kernel", default="search_kernel")
    ap.add_argument("--per", type=float, default=0.0)
    args = ap.parse_args()
    acc = defaultdict(list)
    for d in args.dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per_dispatch = defaultdict(float)
            for r in csv.DictReader(open(f)):
                if args.kernel not in r["Kernel_Name"]:
                    continue
                per_dispatch[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
            for (_, c), v in per_dispatch.items():
                acc[c].append(v)
    for c in sorted(acc):
        v = acc[c]
        m = sum(v) / len(v)
        line = f"{c:28s} dispatches={len(v):3d} mean={m:16.1f}"
        if args.per:
            line += f"  per_unit={m / args.per:10.2f}"
        print(line)


if __name__ == "__main__":
    main()
