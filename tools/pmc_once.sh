#!/bin/bash
# one rocprofv3 --pmc pass over bench.py (1 step), summary printed:  tools/pmc_once.sh TAG "COUNTER ..." [ENV=VAL ...]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; ctrs=$2; shift 2
for kv in "$@"; do export "$kv"; done
OUT=$R/gpurun_out/pmc_$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $OUT/raw -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu > $OUT/bench.json 2> $OUT/err.txt || echo "pmc run failed"
python3 $R/tools/prof_summary.py $OUT/raw "pmc $tag: $ctrs $*" > $OUT/summary.txt 2>&1
rm -rf $OUT/raw
grep -A6 "^## counters" $OUT/summary.txt | cut -c1-600
