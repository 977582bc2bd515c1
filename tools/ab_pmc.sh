#!/bin/bash
# A/B of two builds of the library (exp_lib4.so, exp_lib5.so): instruction counts and cycles of the descriptor kernels
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
cp visual_odometry_amd/libvo_mi355x.so /tmp/good.so
for e in 4 5; do
  cp exp_lib$e.so visual_odometry_amd/libvo_mi355x.so
  rm -rf gpurun_out/ab_$e
  timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES --output-format csv -d gpurun_out/ab_$e -- python3 bench.py --profile-region --steps 2 > gpurun_out/ab_$e.log 2>&1 || { echo "exp $e failed"; tail -3 gpurun_out/ab_$e.log; break; }
  echo "== LIB $e"
  E=$e python3 - <<'PY'
import csv, glob, re, statistics, collections, os
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"gpurun_out/ab_{os.environ['E']}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    if "describe_tiled" not in k: continue
    d = {c: statistics.median(v) for c, v in acc[k].items()}
    print(f"{k[:40]:40s}", "  ".join(f"{c.replace('SQ_INSTS_','')}={v/1e6:.1f}M" for c, v in sorted(d.items())))
PY
done
cp /tmp/good.so visual_odometry_amd/libvo_mi355x.so
