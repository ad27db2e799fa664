#!/bin/bash
# kernel-trace statistics of a complete solve at n = m = $1 by the reference's driver on the engine
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/headline_stats_$1
mkdir -p $O
export HDSDP_DROP_ATTACH=${HDSDP_DROP_ATTACH:-1} HDSDP_MI355X_CALL_STATS=1
# (graph replay off UNDER THE PROFILER: rocprofiler-sdk 7.2's HSA queue write interceptor dereferences a packet pointer outside
# mapped memory beneath a hipGraphLaunch -- a replay of the dual factor's long-lived exec, several hundred replays into the
# solve; frames, disassembly of the faulting instruction and what was ruled out on our side: profiles/r04_a_headline_segv.txt.
# The plain run and the run with HDM_GRAPHS=0 under the profiler both complete.)
export HDM_GRAPHS=${HDM_GRAPHS:-0}    # (the library's default since round 4)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $R/oracle/_ref/headline_solve_mi355x $1 > $O/solve.log 2>&1
python3 $R/tools/prof_summary.py $O/stats "headline_solve_mi355x $1 (kernel-trace --stats)" > $O/summary_stats.txt 2>&1
rm -rf $O/stats
grep -E "M-forming|Optimization time|wall time below" $O/solve.log
head -24 $O/summary_stats.txt | cut -c1-150
