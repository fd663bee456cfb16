"""A/B of library builds on ONE box, interleaved in one process sequence (cdna_hip_programming.md rule 24):
    python scripts/ab_libs.py --config c2 --k 10 --rounds 3 product build/libcph_r2.so ...
Each round runs every library once (a fresh child process per run: a library is loaded once per process), best-of-3
kernel time inside the run; prints the per-library minimum and median over rounds."""
import argparse
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+", help="'product' or a path to a diagnostic build")
    ap.add_argument("--config", default="c2")
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--nq", type=int, default=10000)
    args = ap.parse_args()
    res = {l: [] for l in args.libs}
    for r in range(args.rounds):
        for l in args.libs:
            cmd = [sys.executable, os.path.join(ROOT, "scripts", "phase_timers.py"), "--config", args.config, "--k", str(args.k),
                   "--nq", str(args.nq), "--reps", "5"] + (["--product"] if l == "product" else ["--lib", l])
            out = subprocess.run(cmd, capture_output=True, text=True)
            lines = [x for x in out.stdout.splitlines() if x.startswith("{")]
            if not lines:
                print(l, "FAILED", out.stderr[-300:])
                continue
            res[l].append(json.loads(lines[-1])["best_kernel_us"])
    for l in args.libs:
        v = res[l]
        if v:
            print(f"{args.config:7s} {os.path.basename(l):28s} min {min(v):10.1f} us   median {statistics.median(v):10.1f} us   runs {v}")


if __name__ == "__main__":
    main()
