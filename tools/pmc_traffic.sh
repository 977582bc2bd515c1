#!/bin/bash
# HBM-side traffic and utilisation per kernel launch of the headline workload, every kernel alone on the main stream
# (`bench.py --profile-region`): FETCH_SIZE and WRITE_SIZE in separate --pmc passes (MI355X_MICROARCH.md, HBM / rocprofv3
# section), then an instruction-mix pass.  Also the --kernel-trace --stats summary of the same command.
# Writes gpurun_out/pmc_{fetch,write,util}/, gpurun_out/region_prof/, gpurun_out/hbm_traffic.json, gpurun_out/pmc_util_summary.txt
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
STEPS=${STEPS:-3}
rm -rf gpurun_out/region_prof
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/region_prof -- python3 bench.py --profile-region --steps $STEPS > gpurun_out/region_line.json 2> gpurun_out/region_prof.log || { echo "kernel-trace pass failed"; tail -5 gpurun_out/region_prof.log; exit 1; }
echo "kernel-trace pass done"
for pass in fetch:FETCH_SIZE write:WRITE_SIZE "util:GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_BUSY_CYCLES"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rm -rf gpurun_out/pmc_$name
  timeout -s KILL 240 rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/pmc_$name -- python3 bench.py --profile-region --steps $STEPS > gpurun_out/pmc_$name.log 2>&1 || { echo "pass $name failed"; tail -5 gpurun_out/pmc_$name.log; exit 1; }
  echo "pass $name done"
done
python3 - <<'PY'
import csv, glob, json, re, statistics, collections
def load(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
            acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc
fe, wr, ut = load("pmc_fetch"), load("pmc_write"), load("pmc_util")
out = {"_note": "per launch of `bench.py --profile-region` (64 pairs/step, every kernel alone on the main stream), median over launches; FETCH_SIZE and WRITE_SIZE from separate rocprofv3 --pmc passes, KB; hbm_bytes = (2*FETCH + WRITE)*1024 (gfx950 correction for wide reads; uncalibrated for the gathers of these kernels)"}
for k in sorted(fe):
    f = statistics.median(fe[k]["FETCH_SIZE"]); w = statistics.median(wr[k]["WRITE_SIZE"]) if k in wr and wr[k]["WRITE_SIZE"] else 0.0
    out[k] = {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "hbm_bytes_per_launch": (2 * f + w) * 1024}
json.dump(out, open("gpurun_out/hbm_traffic.json", "w"), indent=1)
with open("gpurun_out/pmc_util_summary.txt", "w") as fo:
    for k in sorted(ut):
        d = {c: statistics.median(v) for c, v in ut[k].items()}
        cyc = d.get("GRBM_GUI_ACTIVE", 0) / 8.0
        line = f"{k[:70]:70s} cycles/XCD {cyc:12.0f}  VALU busy {d.get('SQ_INSTS_VALU', 0) * 4.4 / max(cyc * 1024, 1):5.2f}  " + "  ".join(f"{c}={v:.0f}" for c, v in sorted(d.items()))
        fo.write(line + "\n")
        if "match" in k or "describe_tiled" in k or "hessian" in k or "integral" in k: print(line)
PY
