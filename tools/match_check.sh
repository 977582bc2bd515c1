#!/bin/bash
# matcher work: parity tests (matcher, pipeline, golden), then the profiled bench; prints per-kernel times
set -o pipefail
timeout -k 10 600 python -u -m pytest tests/test_gpu_match.py tests/test_gpu_pipeline.py tests/test_gpu_golden.py -x -q --timeout 180 --timeout-method thread > gpurun_out/match_test.log 2>&1
rc=$?; tail -5 gpurun_out/match_test.log; test $rc -eq 0 || exit $rc
rm -rf gpurun_out/prof_e
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_e -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_prof_e.log 2>&1 || { tail -5 gpurun_out/bench_prof_e.log; exit 1; }
python3 - <<'PY'
import csv,glob,json,os
f=max(glob.glob("gpurun_out/prof_e/*/*kernel_stats.csv"),key=os.path.getmtime)
for r in list(csv.DictReader(open(f))):
    if "match" in r["Name"]: print(r["Name"][:60], r["Calls"], round(float(r["AverageNs"])/1e6,3), r["Percentage"])
for l in open("gpurun_out/bench_prof_e.log"):
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], {k:v for k,v in d["stage_ms_per_step"].items() if "match" in k}, d.get("match_exact_scans_per_step"))
PY
