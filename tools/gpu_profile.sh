#!/bin/bash
# run on the GPU box from the repo root: kernel-trace stats + PMC passes of bench.py (each in its own run)
#   tools/gpu_profile.sh <tag> [bench.py arguments, e.g. --m 8000]     (the program stands directly after "--": no env, no shell)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_${1:-r02}
shift
BARGS="$@"
NM=$(python3 -c "
import sys
a = sys.argv[1:]
g = lambda k, d: a[a.index(k) + 1] if k in a else d
print(g('--n', '2000') + 'x' + g('--m', '2000'))" $BARGS)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu $BARGS > $OUT/stats_bench.json 2> $OUT/stats.err || echo "stats run failed"
python3 $R/tools/prof_summary.py $OUT/stats "bench.py --steps 2 --warmup 1 $BARGS (kernel-trace --stats)" > $OUT/summary_stats.txt 2>&1
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pmc_$tag -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu $BARGS > $OUT/pmc_$tag.json 2> $OUT/pmc_$tag.err || echo "pmc $tag failed"
  python3 $R/tools/prof_summary.py $OUT/pmc_$tag "pmc: $pass" > $OUT/summary_pmc_$tag.txt 2>&1
done
# HBM bytes per launch for bench.py's roofline.traffic (the two passes above), then drop the raw CSVs (hundreds of MB)
python3 $R/tools/prof_summary.py --traffic --nm $NM $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE > $OUT/traffic.json 2> $OUT/traffic.err
rm -rf $OUT/pmc_*/
rm -rf $OUT/stats/*/*kernel_trace.csv $OUT/stats/*/*agent* 2>/dev/null
ls -la $OUT
