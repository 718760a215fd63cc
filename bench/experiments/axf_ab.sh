#!/bin/bash
# interleaved A/B of BN apply-on-load thresholds (stagings per activation element above which the BN is materialised)
mkdir -p gpurun_out/r2
for red in 0 4 8 12 24 1e9 0 4 12; do
  DPA_AXF_MAX_RED=$red timeout -k 10 200 python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('max_red=$red', d['value'], d['ms_per_step'], d['config']['peak_mem_gb'])" | tee -a gpurun_out/r2/axf_ab.txt || exit 1
done
