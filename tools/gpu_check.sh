#!/bin/bash
# GPU-side check used during kernel work: detect parity tests, then a profiled bench run; prints per-kernel times.
set -o pipefail
timeout -k 10 600 python -u -m pytest tests/test_gpu_detect.py -x -q -m gpu --timeout 120 --timeout-method thread > gpurun_out/t_detect.log 2>&1
rc=$?; tail -3 gpurun_out/t_detect.log; test $rc -eq 0 || exit $rc
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/prof_e
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_e -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_prof_e.log 2>&1 || exit $?
python3 - <<'PY'
import csv,glob,json,os
f=max(glob.glob("gpurun_out/prof_e/*/*kernel_stats.csv"),key=os.path.getmtime)
for r in list(csv.DictReader(open(f)))[:10]: print(r["Name"][:60], r["Calls"], round(float(r["AverageNs"])/1e6,3), r["Percentage"])
for l in open("gpurun_out/bench_prof_e.log"):
    if l.startswith("{"): d=json.loads(l); print(d["value"], d["stage_ms_per_step"])
PY
