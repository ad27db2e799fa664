#!/bin/bash
# sample power / clocks / temperature of the device every 0.5 s while bench.py runs (rocm-smi needs no privileges to read)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/power_${1:-x}
mkdir -p $O
( while true; do rocm-smi --showpower --showclocks --showtemp --csv 2>/dev/null | tail -n +2 | head -2 | tr '\n' ' '; echo; sleep 0.5; done ) > $O/smi.csv &
SMI=$!
python3 $R/bench.py --steps 20 --warmup 3 --no-cpu > $O/bench.json 2> $O/bench.err
kill $SMI
rocm-smi --showpower --showclocks --showtemp --csv > $O/smi_header.txt 2>&1
rocm-smi -a > $O/smi_all.txt 2>&1
wc -l $O/smi.csv
