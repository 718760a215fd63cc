#!/bin/bash
# MFMA busy of the kernels one script launches: SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs
# usage: pmc_mfma.sh <script.py> <tag> <kernel name filter> <title>      (conv8: conv8_pmc.py pmc_conv8 conv8 '...')
set -e
script=$1; export TAG=$2; export FILTER=$3; export TITLE=$4
R=$PWD; export PYTHONPATH=$R; cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r3
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/r3/$TAG -- python3 $R/bench/experiments/$script > $R/gpurun_out/r3/$TAG.log 2>&1
cd $R
python3 - <<'PY' | tee gpurun_out/r3/$TAG.md
import csv, glob, collections, os
f = glob.glob("gpurun_out/r3/%s/**/*counter_collection.csv" % os.environ["TAG"], recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("dpa::", "").split("(")[0][:60] + " grid=" + r.get("Grid_Size", "?")
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": n[k] += 1
print("# rocprofv3 --pmc, %s; per dispatch means\n" % os.environ["TITLE"])
print("| kernel | dispatches | MFMA busy cycles / SIMD | GRBM_GUI_ACTIVE / XCD | MFMA busy % | SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES | LDS insts |")
print("|---|---|---|---|---|---|---|")
for k, c in acc.items():
    if os.environ["FILTER"] not in k or not n[k]: continue
    d = n[k]; m = c["SQ_VALU_MFMA_BUSY_CYCLES"] / d / 1024; g = c["GRBM_GUI_ACTIVE"] / d / 8
    print("| `%s` | %d | %.0f | %.0f | %.1f | %.0f %% | %.2fM |" % (k, d, m, g, 100 * m / g, 100 * c["SQ_WAIT_INST_ANY"] / max(c["SQ_WAVE_CYCLES"], 1), c["SQ_INSTS_LDS"] / d / 1e6))
PY
rm -rf gpurun_out/r3/$TAG
