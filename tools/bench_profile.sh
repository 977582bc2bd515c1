#!/bin/bash
# Round bench + profile: the default bench.py line, then rocprofv3 --kernel-trace --stats of the same command.
# Outputs under gpurun_out/: bench_line.json, bench_prof/ (csv), bench_prof_line.json
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python3 bench.py > gpurun_out/bench_line.json 2> gpurun_out/bench_line.err || { tail -5 gpurun_out/bench_line.err; exit 1; }
rm -rf gpurun_out/bench_prof
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bench_prof -- python3 bench.py > gpurun_out/bench_prof_line.json 2> gpurun_out/bench_prof.err || { tail -5 gpurun_out/bench_prof.err; exit 1; }
python3 - <<'PY'
import json
for f in ("gpurun_out/bench_line.json", "gpurun_out/bench_prof_line.json"):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l)
            print(f, d["value"], d["ms_per_step"], d.get("resident", {}).get("value"), d.get("cpu_baseline"), d["roofline"]["kernel"], d["roofline"]["frac"], d.get("match_exact_scans_per_step"))
PY
