#!/bin/bash
# kernel table of one bench configuration: r3_prof.sh <tag> <title> [bench.py args...]
set -o pipefail
tag=$1; title=$2; shift 2
O=gpurun_out/r3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -- python bench.py --steps 6 --warmup 3 "$@" > $O/prof_$tag.log 2>&1 || { tail -5 $O/prof_$tag.log; exit 1; }
python scripts/summarize_rocprof.py $O/prof_$tag 9 "$title" ${ROWS:-30} > $O/$tag.md && head -60 $O/$tag.md
if [ -n "$GRID" ]; then python scripts/summarize_rocprof_grid.py $O/prof_$tag 9 "$GRID" "$title: $GRID by grid" > $O/${tag}_grid.md; fi
