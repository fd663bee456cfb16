"""Static, itemised instruction count of one kernel by source region and instruction class.

    hipcc <product flags> -gline-tables-only -save-temps ... cphnsw_mi355x.hip      (scripts/isa_breakdown.py --build DIR)
    python scripts/isa_breakdown.py DIR/cphnsw_mi355x-hip-amdgcn-amd-amdhsa-gfx950.s --kernel 'search_kernelILi4ELi128'

Every machine instruction of the kernel is attributed to the source line of its innermost inlined frame (the `.loc`
directive in front of it) and from there to a named region: the functions of device_search.h / device_fastscan.h by
their line ranges, the kernel body by the phase markers (CPH_TICK sites).  Classes: VALU, SALU split into exec-mask
bookkeeping / branches / compares / address and integer arithmetic / moves / bit ops, lane broadcasts
(v_readlane / v_readfirstlane / v_writelane: vector-issued, scalar results), SMEM, VMEM, LDS, waitcnt and the rest.
The count is STATIC (code size per region): regions that run once per expansion read directly as instructions per
expansion; loops and cold paths are marked as such in the committed table (profiles/r3_search_isa_breakdown.md).
"""
import argparse
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(out_dir):
    sys.path.insert(0, os.path.join(ROOT, "rabitq-ann-search_amd"))
    from cphnsw_mi355x import build as b
    os.makedirs(out_dir, exist_ok=True)
    cmd = [b._hipcc()] + b.FLAGS + ["-gline-tables-only", "-save-temps", os.path.join(b.CSRC, b.SOURCES[0]), "-o",
                                    os.path.join(out_dir, "lib.so"), "-lpthread"]
    subprocess.check_call(cmd, cwd=out_dir, stderr=subprocess.DEVNULL)
    print(os.path.join(out_dir, "cphnsw_mi355x-hip-amdgcn-amd-amdhsa-gfx950.s"))


def classify(op, operands):
    if op.startswith("v_readlane") or op.startswith("v_readfirstlane") or op.startswith("v_writelane"):
        return "lane<->scalar"
    if op.startswith("v_"):
        return "VALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM"
    if op.startswith(("s_load", "s_buffer_load", "s_store", "s_memtime", "s_memrealtime", "s_dcache")):
        return "SMEM"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_call", "s_endpgm")):
        return "SALU branch"
    if op.startswith(("s_nop", "s_sleep", "s_barrier", "s_setprio", "s_sethalt", "s_sendmsg", "s_trap", "s_code_end", "s_inst_prefetch")):
        return "nop/barrier"
    if op.startswith("s_"):
        if "exec" in operands or "saveexec" in op:
            return "SALU exec mask"
        if op.startswith(("s_cmp", "s_bitcmp", "s_cselect", "s_cmov")):
            return "SALU compare/select"
        if op.startswith(("s_mov", "s_movk")):
            return "SALU move"
        if op.startswith(("s_add", "s_sub", "s_mul", "s_lshl", "s_lshr", "s_ashr", "s_min", "s_max", "s_abs")):
            return "SALU int arith"
        return "SALU bit/logic"
    return "other"


def load_regions(csrc):
    """Named line ranges.  Functions of device_search.h / device_fastscan.h are found by their signatures; the kernel
    body is cut at the CPH_TICK markers."""
    regions = {}

    def scan(fname, patterns):
        lines = open(os.path.join(csrc, fname)).read().split("\n")
        starts = []
        for i, l in enumerate(lines, 1):
            for name, pat in patterns:
                if re.search(pat, l):
                    starts.append((i, name))
        return lines, sorted(starts)

    lines, starts = scan("device_search.h", [
        ("cold_arg", r"^__device__ __forceinline__ T cold_arg"),
        ("lane-0 heaps (nn_sift/adjust/sort)", r"^__device__ __forceinline__ void nn_sift_up"),
        ("beam_spill_off", r"^__device__ __forceinline__ uint32_t beam_spill_off"),
        ("Beam / NnLds accessors", r"^struct Beam \{"),
        ("heap_pop_wave", r"^__device__ __forceinline__ void heap_pop_wave"),
        ("heap_push_wave", r"^__device__ __forceinline__ void heap_push_wave"),
        ("beam_pop_hybrid", r"^__device__ __forceinline__ void beam_pop_hybrid"),
        ("beam_push_hybrid", r"^__device__ __forceinline__ void beam_push_hybrid"),
        ("nn_push_wave", r"^__device__ __forceinline__ void nn_push_wave"),
        ("bcast / lds_dma helpers", r"^__device__ __forceinline__ uint32_t bcast_u32"),
        ("kernel: setup per launch / per query", r"^__global__ __launch_bounds__.*search_kernel"),
        ("kernel: top, termination tests", r"termination tests on the beam's top"),
        ("kernel: load issue, touch, pop call, retire, probe issue", r"this expansion's loads, then the estimated-set probe"),
        ("kernel: exact distance of the popped node, nn push", r"exact distance of the popped node; nn.push"),
        ("kernel: probe result, marking, slack, all-seen exit", r"estimated set: result of the probe issued above"),
        ("kernel: FastScan sums + estimator call sites", r"^\s*LaneEst v;"),
        ("kernel: candidate mask, id log", r"const bool warmup = nn_sz < k;"),
        ("kernel: speculative exact L2 of candidates", r"speculative exact L2 of the candidates"),
        ("kernel: replay, all-lanes path (no rerank)", r"serial replay of the neighbour loop"),
        ("kernel: replay, wave-uniform loop (rerank)", r"past the warm-up the threshold only falls"),
        ("kernel: expansion tail", r"^\s*CPH_TICK\(5\);"),
        ("kernel: results, statistics, bitmap clear", r"---- results \(:276"),
    ])
    for j, (ln, name) in enumerate(starts):
        end = starts[j + 1][0] - 1 if j + 1 < len(starts) else len(lines)
        regions[("device_search.h", ln, end)] = name
    lines, starts = scan("device_fastscan.h", [(m.group(1), None) for m in []])
    return regions


def region_of(regions, fname, line, fastscan_funcs):
    base = os.path.basename(fname)
    if base == "device_search.h":
        for (f, lo, hi), name in regions.items():
            if lo <= line <= hi:
                return name
        return "device_search.h (other)"
    if base == "device_fastscan.h":
        for lo, hi, name in fastscan_funcs:
            if lo <= line <= hi:
                return "fastscan: " + name
        return "fastscan: (other)"
    if base.startswith("amd_") or base.startswith("__clang"):
        return "hip runtime headers (shuffles, ballots, atomics, math)"
    return base


def fastscan_functions(csrc):
    lines = open(os.path.join(csrc, "device_fastscan.h")).read().split("\n")
    out = []
    cur = None
    for i, l in enumerate(lines, 1):
        m = re.match(r"^(?:template.*>\s*)?__device__ __forceinline__ [\w:<>,\s\*&]+?\s+(\w+)\(", l)
        m2 = re.match(r"^\s+__device__ __forceinline__ [\w:<>,\s\*&]+?\s+(\w+)\(", l)
        if m or m2:
            if cur:
                out.append((cur[0], i - 1, cur[1]))
            cur = (i, (m or m2).group(1))
        elif re.match(r"^struct (\w+)", l):
            if cur:
                out.append((cur[0], i - 1, cur[1]))
            cur = (i, "struct " + re.match(r"^struct (\w+)", l).group(1))
    if cur:
        out.append((cur[0], len(lines), cur[1]))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm", nargs="?")
    ap.add_argument("--build", default="")
    ap.add_argument("--kernel", default="search_kernelILi4ELi128")
    ap.add_argument("--lines", action="store_true", help="also list the 40 heaviest source lines")
    args = ap.parse_args()
    if args.build:
        return build(os.path.abspath(args.build))
    csrc = os.path.join(ROOT, "rabitq-ann-search_amd", "csrc")
    regions = load_regions(csrc)
    fs_funcs = fastscan_functions(csrc)
    files = {}
    table = collections.defaultdict(collections.Counter)
    by_line = collections.Counter()
    inside = False
    cur = ("?", 0)
    for raw in open(args.asm, errors="replace"):
        l = raw.strip()
        m = re.match(r'\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', l)
        if m:
            files[int(m.group(1))] = m.group(3)
            continue
        m = re.match(r'\.file\s+(\d+)\s+"([^"]*)"', l)
        if m:
            files[int(m.group(1))] = m.group(2)
            continue
        if not inside:
            if re.match(r"^_ZN3cph\d+%s\w*:" % re.escape(args.kernel), l) or (l.endswith(":") and args.kernel in l and l.startswith("_Z")):
                inside = True
            continue
        if l.startswith(".Lfunc_end") or l.startswith(".section") or l.startswith(".size"):
            break
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", l)
        if m:
            cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
            continue
        if not l or l.startswith((";", ".", "//")) or l.endswith(":"):
            continue
        parts = l.split(None, 1)
        op = parts[0]
        if not re.match(r"^[a-z]", op):
            continue
        operands = parts[1].split(";")[0] if len(parts) > 1 else ""
        cls = classify(op, operands)
        reg = region_of(regions, cur[0], cur[1], fs_funcs)
        table[reg][cls] += 1
        by_line[(os.path.basename(cur[0]), cur[1])] += 1
    classes = ["VALU", "lane<->scalar", "SALU exec mask", "SALU branch", "SALU compare/select", "SALU int arith", "SALU move",
               "SALU bit/logic", "SMEM", "VMEM", "LDS", "waitcnt", "nop/barrier", "other"]
    print("| region | " + " | ".join(classes) + " | total |")
    print("|---|" + "---|" * (len(classes) + 1))
    tot = collections.Counter()
    for reg in sorted(table, key=lambda r: -sum(table[r].values())):
        row = table[reg]
        tot.update(row)
        print(f"| {reg} | " + " | ".join(str(row.get(c, 0)) for c in classes) + f" | {sum(row.values())} |")
    print("| **all** | " + " | ".join(str(tot.get(c, 0)) for c in classes) + f" | {sum(tot.values())} |")
    if args.lines:
        print()
        for (f, ln), c in by_line.most_common(40):
            print(f"{c:5d}  {f}:{ln}")


if __name__ == "__main__":
    main()
