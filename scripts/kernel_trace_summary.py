"""Condenses a `rocprofv3 --kernel-trace --stats --output-format csv` run of bench.py: per kernel name the launches
longer than 50 us (every search enqueues a second, usually empty, re-run launch of a few us that would halve the
average), and for the search kernel the mean of the LAST `--serial N` such launches -- the serialised measurement
region whose HIP-event mean bench.py reports as roofline.kernel_ms.
usage: kernel_trace_summary.py <..._kernel_trace.csv> [--serial N]"""
import csv, sys
path = sys.argv[1]
serial = int(sys.argv[sys.argv.index("--serial") + 1]) if "--serial" in sys.argv else 23
rows = list(csv.DictReader(open(path)))
by = {}
for r in rows:
    name = r.get("Kernel_Name") or r.get("kernel_name")
    st, en = int(r.get("Start_Timestamp") or r["start_timestamp"]), int(r.get("End_Timestamp") or r["end_timestamp"])
    by.setdefault(name, []).append((st, en - st))
print(f"{'kernel':70s} {'launches':>8s} {'>50us':>6s} {'mean_us(>50us)':>15s} {'min_us':>9s} {'max_us':>9s}")
for name, v in sorted(by.items(), key=lambda kv: -sum(d for _, d in kv[1])):
    long_ = [d for _, d in v if d > 50_000]
    if not long_:
        continue
    print(f"{name[:70]:70s} {len(v):8d} {len(long_):6d} {sum(long_) / len(long_) / 1e3:15.1f} {min(long_) / 1e3:9.1f} {max(long_) / 1e3:9.1f}")
for name, v in by.items():
    if "search_kernel" in name:
        long_ = [d for _, d in sorted(v) if d > 50_000]
        tail = long_[-serial:]
        print(f"\n{name}: last {len(tail)} launches > 50 us (the serialised region): mean {sum(tail) / len(tail) / 1e3:.1f} us, "
              f"min {min(tail) / 1e3:.1f}, max {max(tail) / 1e3:.1f}")
