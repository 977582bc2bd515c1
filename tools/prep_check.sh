#!/bin/bash
# N2 check on the GPU box: parity tests, stage timing, kernel trace of tools/prep_bench.py
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_prep.py -x -q --timeout 120 --timeout-method thread > gpurun_out/prep_test.log 2>&1 || { tail -20 gpurun_out/prep_test.log; exit 1; }
tail -1 gpurun_out/prep_test.log
timeout -k 10 200 python -u tools/prep_bench.py > gpurun_out/prep_bench.json 2> gpurun_out/prep_bench.err || { tail -5 gpurun_out/prep_bench.err; exit 1; }
cat gpurun_out/prep_bench.json
rm -rf gpurun_out/prep_prof
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prep_prof -- python3 tools/prep_bench.py --frames 100 --cpu-frames 1 > gpurun_out/prep_prof.log 2>&1 || { tail -5 gpurun_out/prep_prof.log; exit 1; }
python3 - <<'PY'
import csv, glob, os
fs = sorted(glob.glob("gpurun_out/prep_prof/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
for r in csv.DictReader(open(fs[-1])):
    print(r["Name"][:70], r["Calls"], r["AverageNs"])
PY
