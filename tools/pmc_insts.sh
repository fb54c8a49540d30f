#!/bin/bash
# per-kernel instruction counts of a few steps of the metric workload (rocprofv3 --pmc; its own
# run, kernel trace only): tools/pmc_insts.sh <tag>
TAG=${1:-insts}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$ROOT"
mkdir -p gpurun_out/$TAG
rm -rf gpurun_out/$TAG/pmc
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d gpurun_out/$TAG/pmc -o run -- python3 tools/kbench.py --genomes --steps 8 --no-profile > gpurun_out/$TAG/pmc.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/$TAG/pmc/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][:44]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if r["Counter_Name"] == "SQ_WAVES":
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
print("%-46s %8s %9s %9s %8s %8s %8s %8s" % ("kernel (last 6 launches)", "waves", "VALU/w", "SALU/w", "LDS/w", "VMrd/w", "VMwr/w", "us"))
for k, v in sorted(acc.items(), key=lambda kv: -sum(dur[kv[0]][-6:])):
    w = sum(v["SQ_WAVES"][-6:]) / max(len(v["SQ_WAVES"][-6:]), 1)
    if w == 0: continue
    g = lambda c: sum(v[c][-6:]) / max(len(v[c][-6:]), 1) / w
    print("%-46s %8.0f %9.0f %9.0f %8.1f %8.1f %8.1f %8.1f" % (k, w, g("SQ_INSTS_VALU"), g("SQ_INSTS_SALU"), g("SQ_INSTS_LDS"), g("SQ_INSTS_VMEM_RD"), g("SQ_INSTS_VMEM_WR"), sum(dur[k][-6:]) / max(len(dur[k][-6:]), 1)))
PY
