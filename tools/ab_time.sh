#!/bin/bash
# A/B of experimental builds of the library (exp_lib$e.so for e in $LIBS, default "4 5"): stage times of the headline workload (each kernel alone)
cd "$GRAFT_REPO_ROOT" || exit 1
cp visual_odometry_amd/libvo_mi355x.so /tmp/good.so
for e in ${LIBS:-4 5}; do
  cp exp_lib$e.so visual_odometry_amd/libvo_mi355x.so
  echo "== LIB $e"
  timeout -k 10 200 python3 bench.py --profile-region --steps 4 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())['stage_ms_per_step']
print({k:v for k,v in d.items() if 'describe' in k or 'hessian' in k}, 'sum describe', round(sum(v for k,v in d.items() if 'describe_w' in k),3))" || break
done
cp /tmp/good.so visual_odometry_amd/libvo_mi355x.so
