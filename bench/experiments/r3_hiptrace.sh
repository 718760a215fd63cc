#!/bin/bash
O=gpurun_out/r3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for v in plain ddp; do
  arg=""; [ $v = ddp ] && arg="ddp"
  timeout -k 10 240 rocprofv3 --hip-trace --stats --output-format csv -d $O/hiptrace_$v -- python bench/experiments/ddp_host_profile.py alexnet 512 $arg > $O/hiptrace_$v.log 2>&1 || { tail -20 $O/hiptrace_$v.log; exit 1; }
  f=$(find $O/hiptrace_$v -name "*hip_api_stats.csv" | head -1)
  echo "== $v: $f"; head -14 $f | cut -c1-150
done
