#!/bin/bash
# rocprofv3 --kernel-trace --stats of one bench.py configuration -> gpurun_out/<tag>/kernel_stats.csv  (usage: prof_stats.sh <tag> [bench args])
set -e -o pipefail
tag=$1; shift
OUT=gpurun_out/$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline "$@" > $OUT/prof.log 2>&1
cp "$(find $OUT/prof -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats.csv
TR="$(find $OUT/prof -name '*kernel_trace.csv' | head -1)"
python tools/step_summary.py "$TR" $OUT/step_summary.json ${WL:-joint} ${BATCH:-16} ${FRAMES:-500} || true
python tools/chain_timeline.py "$TR" 10 > $OUT/chain_timeline.txt || true
head -3 "$TR" > $OUT/trace_head.txt; python tools/cu_time.py "$TR" 5 40 > $OUT/cu_time.txt || true
rm -rf $OUT/prof
tail -1 $OUT/prof.log
