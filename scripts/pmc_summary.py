"""Summarises rocprofv3 --pmc result databases (rocpd sqlite) per kernel:
    python scripts/pmc_summary.py gpurun_out/pmcA/runc/NNN_results.db ... --kernel search_kernel --per 2629998
prints the mean counter value per dispatch and per unit (--per = e.g. expansions per dispatch)."""
import argparse
import sqlite3
from collections import defaultdict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dbs", nargs="+")
    ap.add_argument("--kernel", default="search_kernel")
    ap.add_argument("--per", type=float, default=0.0)
    args = ap.parse_args()
    for f in args.dbs:
        con = sqlite3.connect(f)
        acc = defaultdict(lambda: defaultdict(float))
        q = "select dispatch_id, counter_name, value from counters_collection where kernel_name like ?"
        for d, c, v in con.execute(q, (f"%{args.kernel}%",)):
            acc[c][d] += v
        for c in sorted(acc):
            vals = list(acc[c].values())
            m = sum(vals) / len(vals)
            line = f"{c:24s} dispatches={len(vals):3d} mean={m:14.4g}"
            if args.per:
                line += f"  per_unit={m / args.per:9.1f}"
            print(line)


if __name__ == "__main__":
    main()
