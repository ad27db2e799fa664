#!/bin/bash
# run on the GPU box from the repo root: one kernel-trace --stats pass of bench.py, summarised ($1 = tag; extra environment from the caller)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/stats_${1:-x}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu > $OUT/bench.json 2> $OUT/stats.err || echo "stats run failed"
python3 $R/tools/prof_summary.py $OUT/stats "bench.py --steps 2 --warmup 1 (kernel-trace --stats)" > $OUT/summary_stats.txt 2>&1
rm -rf $OUT/stats
head -30 $OUT/summary_stats.txt
