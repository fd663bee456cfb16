#!/bin/bash
# One table of the diagnostic / A-B builds of the C-ABI library (git-ignored outputs under build/; they travel to the GPU
# box with the snapshot).  A variant = the product's flags (cphnsw_mi355x/build.py FLAGS) + the defines below; load one with
# CPH_LIB_PATH=$PWD/build/libcph_<variant>.so (bench.py, scripts/phase_timers.py --lib, scripts/ab_libs.py).
#     scripts/ab_build.sh <variant> [<variant> ...]      scripts/ab_build.sh --list
set -e
cd "$(dirname "$0")/.."
declare -A V=(
  [base]=""                                              # the product's own flags (a frozen copy for in-box A/B)
  [timers]="-DCPH_PHASE_TIMERS"                          # per-phase cycle counters of the search kernel (stats[8..15])
  [fine]="-DCPH_PHASE_TIMERS=2"                          # finer buckets along one expansion's dependent chain
  [traffic]="-DCPH_TRAFFIC_STATS"                        # what spilled beams and the estimated-set probe touch
  [launder]="-DCPH_LAUNDER_HOT"                          # hot heap routines recompute their lane arithmetic (fewer VGPRs)
  [nopf]="-DCPH_NO_PROBE_FIRST"                          # D=128: codes always fetched with the ids (the library never probes first)
  [noappend]="-DCPH_NO_SKIP_APPEND"                      # single push onto a spilled beam takes the append path
  [nolaundernn]="-DCPH_NO_LAUNDER_NN"                    # nn_push_wave keeps hoisted lane arithmetic
  [noloopwait]="-DCPH_NO_LOOPHEAD_WAIT"                  # no vmcnt(0) at the head of the expansion loop
  [w5]="-DCPH_SEARCH_WAVES_PER_SIMD_128=5"               # five waves per SIMD (96 VGPRs) for D = 128
  [w7]="-DCPH_SEARCH_WAVES_PER_SIMD_128=7"               # seven waves per SIMD (72 VGPRs)
  [trace]="-DCPH_SEARCH_TRACE"                           # host-side timing of the coalesced cph_search launches (stderr at cph_destroy)
)
if [ "$1" = "--list" ] || [ -z "$1" ]; then for k in "${!V[@]}"; do printf "%-14s %s\n" "$k" "${V[$k]}"; done | sort; exit 0; fi
FLAGS=$(python3 -c "import sys; sys.path.insert(0,'rabitq-ann-search_amd'); from cphnsw_mi355x import build as b; print(' '.join(b.FLAGS))")
mkdir -p build
for v in "$@"; do
  extra="${V[$v]-__none__}"
  if [ "$extra" = "__none__" ]; then
    case "$v" in -D*) extra="$v"; v=$(echo "$v" | tr -c 'A-Za-z0-9\n' '_');; *) echo "unknown variant $v (try --list, or pass -DNAME[=x])"; exit 2;; esac
  fi
  /opt/rocm/bin/hipcc $FLAGS $extra rabitq-ann-search_amd/csrc/cphnsw_mi355x.hip -o build/libcph_$v.so -lpthread 2> build/libcph_$v.log || { tail -20 build/libcph_$v.log; exit 1; }
  echo "build/libcph_$v.so   [$extra]"
done
