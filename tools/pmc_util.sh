#!/bin/bash
# one --pmc pass (instruction mix) of the default bench workload; prints the descriptor / Hessian / matcher kernels
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/pmc_util
timeout -s KILL 240 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d gpurun_out/pmc_util -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_util.log 2>&1 || { tail -5 gpurun_out/pmc_util.log; exit 1; }
python3 - <<'PY'
import csv, glob, re, statistics, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_util/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    if not any(s in k for s in ("describe_tiled", "hessian", "match_bf3", "orientation")): continue
    d = {c: statistics.median(v) for c, v in acc[k].items()}
    print(f"{k[:44]:44s}", "  ".join(f"{c.replace('SQ_INSTS_','')}={v/1e6:.1f}M" for c, v in sorted(d.items())))
PY
