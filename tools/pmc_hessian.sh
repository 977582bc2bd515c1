#!/bin/bash
# where the Hessian kernels' wavefront cycles go: two --pmc passes over `bench.py --profile-region` (every kernel alone on the main stream)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for pass in "a:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "b:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAVES GRBM_GUI_ACTIVE"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rm -rf gpurun_out/pmc_h_$name
  timeout -s KILL 200 rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/pmc_h_$name -- python3 bench.py --profile-region --steps 2 > gpurun_out/pmc_h_$name.log 2>&1 || { echo "pass $name failed"; tail -5 gpurun_out/pmc_h_$name.log; exit 1; }
  echo "pass $name done"
done
python3 - <<'PY'
import csv, glob, re, statistics, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pmc_h_a", "pmc_h_b"):
    for f in glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
            acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("gpurun_out/pmc_hessian.txt", "w") as fo:
    for k in sorted(acc):
        if not any(s in k for s in ("hessian", "describe_tiled", "orientation")): continue
        d = {c: statistics.median(v) for c, v in acc[k].items()}
        line = f"{k[:44]:44s} " + "  ".join(f"{c}={v:.4g}" for c, v in sorted(d.items()))
        fo.write(line + "\n"); print(line)
PY
