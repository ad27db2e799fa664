#!/bin/bash
# copy the summaries of one tools/gpu_profile.sh run from gpurun_out/ (scratch) into profiles/ (tracked):
#   tools/keep_profile.sh r02a r02_a [bench.json]
src=gpurun_out/prof_$1; dst=profiles/$2
for f in $src/summary_*.txt; do cp $f ${dst}_$(basename $f); done
cp $src/traffic.json ${dst}_traffic.json
cp $src/stats_bench.json ${dst}_bench_under_rocprof.json
[ -n "$3" ] && cp $3 ${dst}_bench.json
ls ${dst}_*
