"""Summarises rocprofv3 --pmc result databases (rocpd sqlite) per kernel:
    python scripts/pmc_summary.py a_results.db b_results.db ... --kernel search_kernel [--per N]
prints the mean counter value per dispatch (and per unit: --per = e.g. expansions per dispatch).
With --stats-log (the JSON line scripts/phase_timers.py printed under the profiler) and --json it also
writes the record bench.py reads for `roofline.traffic`: HBM bytes = 2 x FETCH_SIZE (gfx950 rule for wide
coalesced reads, MI355X_MICROARCH.md section HBM; calibrated on the stream kernel) + WRITE_SIZE, in KB as
rocprofv3 reports them, over the algorithmic bytes of the same dispatches."""
import argparse
import json
import sqlite3
from collections import defaultdict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dbs", nargs="+")
    ap.add_argument("--kernel", default="search_kernel")
    ap.add_argument("--per", type=float, default=0.0)
    ap.add_argument("--stats-log", default="")
    ap.add_argument("--json", default="")
    args = ap.parse_args()
    stats = None
    if args.stats_log:
        for line in open(args.stats_log, errors="replace"):
            line = line.strip()
            if line.startswith("{") and "stats" in line:
                stats = json.loads(line)
    per = args.per or (stats["stats"]["expansions"] if stats else 0.0)
    means = {}
    for f in args.dbs:
        con = sqlite3.connect(f)
        acc = defaultdict(lambda: defaultdict(float))
        q = "select dispatch_id, counter_name, value from counters_collection where kernel_name like ?"
        for d, c, v in con.execute(q, (f"%{args.kernel}%",)):
            acc[c][d] += v
        for c in sorted(acc):
            vals = sorted(acc[c].values())
            # every batch is followed by the (normally empty) overflow re-run launch of the same kernel:
            # keep the real launches only -- identical batches, so anything below 5 % of the largest is one of those
            vals = [v for v in vals if v >= 0.05 * vals[-1]] or vals
            m = sum(vals) / len(vals)
            means[c] = m
            line = f"{c:24s} dispatches={len(vals):3d} mean={m:14.5g}"
            if per:
                line += f"  per_expansion={m / per:9.2f}"
            print(line)
    if args.json and stats and "FETCH_SIZE" in means and "WRITE_SIZE" in means:
        st = stats["stats"]
        import bench_shapes
        bits = stats.get("bits", 0)
        alg = st["expansions"] * 32 * bench_shapes.bytes_per_dist(stats["config"], bits) + st["exact_l2"] * bench_shapes.bytes_per_exact(stats["config"])
        hbm = 2.0 * means["FETCH_SIZE"] * 1024.0 + means["WRITE_SIZE"] * 1024.0
        rec = {"config": stats["config"], "k": stats["k"], "bits": bits or bench_shapes._cfg(stats["config"])["bits"], "nq": stats["nq"],
               "kernel_us": st.get("kernel_us"), "expansions": st["expansions"],
               "exact_l2": st["exact_l2"], "algorithmic_bytes": alg, "fetch_size_kb": means["FETCH_SIZE"],
               "write_size_kb": means["WRITE_SIZE"], "hbm_bytes": hbm, "hbm_bytes_over_algorithmic": hbm / alg,
               "per_expansion": {c: means[c] / st["expansions"] for c in means}}
        json.dump(rec, open(args.json, "w"), indent=1)
        print("traffic / algorithmic =", round(hbm / alg, 4))


if __name__ == "__main__":
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    main()
