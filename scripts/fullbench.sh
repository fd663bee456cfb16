#!/bin/bash
# the driver's default command, as the driver runs it (fresh box: builds every index), with its wall time
export TMPDIR=/tmp
O=${1:-gpurun_out/fullbench}; mkdir -p $O
( while true; do sleep 60; echo "[keepalive] $(date +%T)" >> $O/progress.log; done ) &
KA=$!
t0=$(date +%s)
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; rc=$?
kill $KA
echo "rc=$rc wall=$(( $(date +%s) - t0 )) s"
python3 - <<PY
import json
j=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
r=j["roofline"]
print("value",round(j["value"]),"ms/step",round(j["ms_per_step"],4),"kernel_ms",round(r["kernel_ms"],4),"frac",round(r["frac"],4),"moved_frac",r.get("moved_frac"),"full_queue",r["full_queue"] and {k:(round(v,3) if isinstance(v,float) else v) for k,v in r["full_queue"].items() if k!="measured"})
print("qps_serial",round(j["qps_serial"]),"host_api",round(j["qps_host_api"]),"stream frac",round(j["fastscan_stream"]["roofline"]["frac"],4))
cb=j.get("cpu_baseline",{}); print("cpu",cb.get("value"),cb.get("parity_vs_reference"),cb.get("parity_vs_oracle_counters"))
for k in ("qps_at_recall_gate","qps_at_recall_gate_4bit"):
    g=j.get(k,{}); print(k,{x:g.get(x) for x in ("value","k","bits","recall_target_met","kernel_frac_of_hbm_peak","kernel_moved_frac_of_hbm_peak","reference_qps","parity_vs_reference","parity_vs_oracle_counters","error")}, g.get("recall_at_10"))
for k,v in j.get("legs",{}).items(): print(k,{x:v.get(x) for x in ("value","roofline","parity_vs_reference","parity_vs_oracle","parity_vs_oracle_counters","index_build_s","error","started_at_s")})
print("failed",j["legs_failed"],"skipped",j["legs_skipped"],"parity_failures",j["parity_failures"],"run_s",j["run_s"])
PY
