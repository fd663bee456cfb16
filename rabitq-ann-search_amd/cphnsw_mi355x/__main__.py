"""`python -m cphnsw_mi355x --config configs/benchmark.yaml` — same YAML keys and stdout JSON events as
the reference's `python -m cphnsw` (cphnsw/__main__.py:17-66, configs/benchmark.yaml)."""
import argparse
import json
from pathlib import Path

import yaml

from .datasets import ALL_DATASETS
from .eval import run_benchmark


def main(argv=None):
    ap = argparse.ArgumentParser(prog="cphnsw_mi355x", description="Run the CP-HNSW benchmark on MI355X.")
    ap.add_argument("--config", type=Path, required=True, help="Path to benchmark config YAML.")
    args = ap.parse_args(argv)
    cfg = yaml.safe_load(args.config.read_text())
    out_dir = Path(cfg["run"]["output_dir"])
    names = ALL_DATASETS if cfg["data"]["dataset"] == "all" else [cfg["data"]["dataset"]]
    outputs = []
    for name in names:
        print(json.dumps({"event": "benchmark_start", "dataset": name}), flush=True)
        outputs.append(run_benchmark(name, Path(cfg["data"]["base_dir"]), cfg["eval"]["k"], cfg["eval"]["n_runs"],
                                     out_dir))
    for o in outputs:
        for r in o["results"]:
            print(json.dumps({"event": "summary", "dataset": o["metadata"]["dataset"], "algorithm": r["algorithm"],
                              "build_time_min": round(r["build_time_s"] / 60.0, 4),
                              "memory_gib": round(r["memory_mb"] / 1024.0, 4),
                              "recall_at_10": r["recall_at_10"], "qps": r["qps"]}), flush=True)


if __name__ == "__main__":
    main()
