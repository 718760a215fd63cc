#!/bin/bash
# HBM bytes (FETCH_SIZE / WRITE_SIZE, KiB... units of the counter) per kernel of one script: pmc_bytes.sh <script.py> <tag>
set -e
script=$1; tag=$2
R=$PWD; export PYTHONPATH=$R; cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r3
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/r3/${tag}_rd -- python3 $R/bench/experiments/$script > $R/gpurun_out/r3/${tag}_rd.log 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/r3/${tag}_wr -- python3 $R/bench/experiments/$script > $R/gpurun_out/r3/${tag}_wr.log 2>&1
cd $R
python3 - <<PY | tee gpurun_out/r3/$tag.md
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for sub in ("rd", "wr"):
    f = glob.glob("gpurun_out/r3/${tag}_%s/**/*counter_collection.csv" % sub, recursive=True)[0]
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("dpa::", "").split("(")[0][:70]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
print("| kernel | FETCH_SIZE / dispatch | WRITE_SIZE / dispatch | TCC hit rate | GRBM_GUI_ACTIVE / XCD |")
print("|---|---|---|---|---|")
for k, c in acc.items():
    d = lambda name: c[name] / max(n[k][name], 1)
    hit = c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1)
    print("| \`%s\` | %.1f MB | %.1f MB | %.0f %% | %.0f |" % (k, d("FETCH_SIZE") / 1024, d("WRITE_SIZE") / 1024, 100 * hit, d("GRBM_GUI_ACTIVE") / 8))
PY
rm -rf gpurun_out/r3/${tag}_rd gpurun_out/r3/${tag}_wr
