"""Per-phase cycle counts of the search kernel (a -DCPH_PHASE_TIMERS build, loaded through
CPH_LIB_PATH).  Build the instrumented library first, here or on the box:
    python scripts/phase_timers.py --build
then on the GPU box:  python scripts/phase_timers.py [--index /tmp/cph_bench/bench_n1000000_b4.idx]
The library prints one "[phase cycles]" line per search to stderr."""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "build", "libcph_timers.so")


def build(extra, timers=True, out=LIB):
    sys.path.insert(0, os.path.join(ROOT, "rabitq-ann-search_amd"))
    from cphnsw_mi355x import build as b
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = [b._hipcc()] + b.FLAGS + (["-DCPH_PHASE_TIMERS"] if timers else []) + extra + [os.path.join(b.CSRC, s) for s in b.SOURCES] + ["-o", out, "-lpthread"]
    subprocess.check_call(cmd)
    print(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--flag", action="append", default=[])
    ap.add_argument("--no-timers", action="store_true")
    ap.add_argument("--lib", default=LIB, help="library to build / load")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--config", default="c2")
    ap.add_argument("--index", default="", help="index file (default: the bench cache of --config)")
    ap.add_argument("--nq", type=int, default=10000)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--bits", type=int, default=0, help="0 = the config's bit width (bench.py --bits)")
    ap.add_argument("--product", action="store_true", help="run the shipped library (no timers), e.g. under rocprofv3 --pmc")
    args = ap.parse_args()
    if args.build:
        return build(args.flag, not args.no_timers, os.path.abspath(args.lib))
    if not args.product:
        os.environ["CPH_LIB_PATH"] = os.path.abspath(args.lib)
    sys.path.insert(0, os.path.join(ROOT, "rabitq-ann-search_amd"))
    sys.path.insert(0, ROOT)
    import torch
    import bench
    from cphnsw_mi355x import CPIndex
    cfg = dict(bench.CONFIGS[args.config])
    if args.bits:
        cfg["bits"] = args.bits

    class _A:
        workdir = os.environ.get("CPH_BENCH_DIR", "/tmp/cph_bench")
        config = args.config
    if not args.index:
        args.index = bench.index_path(_A, cfg, cfg["n"])
    if not os.path.exists(args.index):
        raise SystemExit(f"{args.index} missing: run bench.py --config {args.config} once to build it")
    Q = bench.make_queries(cfg, cfg["n"], args.nq)
    idx = CPIndex(dim=cfg["dim"], bits=cfg["bits"])
    idx.load(args.index)
    q = torch.from_numpy(Q).cuda()
    best = 1e9
    for _ in range(args.reps):
        idx.search_batch_device(q, args.k)
        torch.cuda.synchronize()
        best = min(best, idx.last_search_stats()["kernel_us"])
    import json
    print(json.dumps({"lib": "product" if args.product else os.path.basename(args.lib), "config": args.config, "bits": cfg["bits"], "nq": args.nq, "k": args.k,
                      "best_kernel_us": best, "stats": idx.last_search_stats()}))


if __name__ == "__main__":
    main()
