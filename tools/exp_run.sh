#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
cp visual_odometry_amd/libvo_mi355x.so /tmp/good.so
for e in 4 5; do
  cp exp_lib$e.so visual_odometry_amd/libvo_mi355x.so
  rm -rf gpurun_out/prof_x
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_x -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/exp_$e.log 2>&1 || { echo "exp $e failed"; tail -3 gpurun_out/exp_$e.log; break; }
  echo "== EXP $e"
  python3 - <<'PY'
import csv,glob,os
f=max(glob.glob("gpurun_out/prof_x/*/*kernel_stats.csv"),key=os.path.getmtime)
for r in csv.DictReader(open(f)):
    if os.environ.get("VO_EXP_KERNEL", "describe_tiled") in r["Name"]: print("  ", r["Name"][:46], round(float(r["AverageNs"])/1e6,3))
PY
done
cp /tmp/good.so visual_odometry_amd/libvo_mi355x.so
