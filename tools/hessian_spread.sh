#!/bin/bash
# Per-launch durations of the two Hessian + NMS kernels over back-to-back steps of `bench.py --profile-region` (every kernel
# alone on the main stream), with the kernel that ran before each launch and the idle gap to it: what the 2.5 - 3.6 ms spread of
# k_hessian_nms_lds in profiles/r02_f_region_kernel_stats.csv comes from.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/hess_trace
timeout -s KILL 240 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/hess_trace -- python3 bench.py --profile-region --steps ${STEPS:-8} > gpurun_out/hess_trace_line.json 2> gpurun_out/hess_trace.log || { tail -5 gpurun_out/hess_trace.log; exit 1; }
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/hess_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
prev = None
out = []
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if "hessian_nms" in n:
        out.append(f"{n[:34]:34s} {1e-6*(e-s):7.3f} ms   after {prev[0][:30]:30s} gap {1e-3*(s-prev[1]):8.1f} us   grid {r.get('Grid_Size','?')} wg {r.get('Workgroup_Size','?')}")
    prev = (n, e)
print("\n".join(out))
open("gpurun_out/hessian_spread.txt", "w").write("\n".join(out) + "\n")
PY
