#!/bin/bash
# issue mix of the attention kernels: pmc_attn.sh  (writes gpurun_out/r3/pmc_attn.md)
set -e
R=$PWD; export PYTHONPATH=$R; cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r3
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/r3/pmc_attn1 -- python3 $R/bench/experiments/attn_pmc.py > $R/gpurun_out/r3/pmc_attn1.log 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/r3/pmc_attn2 -- python3 $R/bench/experiments/attn_pmc.py > $R/gpurun_out/r3/pmc_attn2.log 2>&1
cd $R
python3 - <<'PY' | tee gpurun_out/r3/${PMC_OUT:-pmc_attn}.md
import csv, glob, collections
import os
print("# rocprofv3 --pmc, fused attention, shape B,S,H,D = %s (windows %s); per dispatch means\n" % (os.environ.get("ATTN_SHAPE", "128,197,12,64"), os.environ.get("ATTN_WINDOWS", "0")))
for tag in ("pmc_attn1", "pmc_attn2"):
    f = glob.glob("gpurun_out/r3/%s/**/*counter_collection.csv" % tag, recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    names = []
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("dpa::", "").split("(")[0][:60]
        if "attn" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] not in names: names.append(r["Counter_Name"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE": n[k] += 1
    print("| kernel | " + " | ".join(names) + " |")
    print("|---|" + "---|" * len(names))
    for k, c in acc.items():
        print("| `%s` | " % k + " | ".join("%.3g" % (c[m] / max(n[k], 1)) for m in names) + " |")
    print()
PY
rm -rf gpurun_out/r3/pmc_attn1 gpurun_out/r3/pmc_attn2
