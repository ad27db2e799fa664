#!/bin/bash
# The reference's own driver on the engine (oracle/_ref/sdpasolve_mi355x): one line per instance and mode.
# mode 0 = the reference's CPU cones on the engine's Schur operator / linear systems; mode 1 = the engine's cones attached.
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=$(mktemp -d)
python3 $R/tools/synth_sdpa.py 120 120 $T/syn120.dat-s
python3 $R/tools/synth_sdpa.py 200 200 $T/syn200.dat-s
python3 $R/tools/synth_sdpa.py 400 400 $T/syn400.dat-s
python3 $R/tools/synth_sdpa.py 30 100 $T/syn30x100.dat-s
python3 $R/tools/blocks_sdpa.py $T/blockslp.dat-s lp
for f in theta1 mcp100 gpp100 truss1; do cp $R/tests/golden/$f.dat-s $T/; done
printf "%-12s %-4s %-24s %-18s %-8s %s\n" instance mode status dObj iters "optimisation time"
for f in theta1 mcp100 gpp100 truss1 blockslp syn30x100 syn120 syn200 syn400; do
  for a in 0 1; do
    if [ $a = 0 ] && { [ $f = syn200 ] || [ $f = syn400 ]; }; then continue; fi
    out=$(HDSDP_DROP_ATTACH=$a timeout -k 10 900 $R/oracle/_ref/sdpasolve_mi355x $T/$f.dat-s 2>&1)
    st=$(echo "$out" | grep "SDP Status" | sed 's/SDP Status: //')
    dobj=$(echo "$out" | grep "dObj" | tail -1 | awk '{print $2}')
    it=$(echo "$out" | grep -E "^ +[0-9]+ +[-+][0-9]" | tail -1 | awk '{print $1}')
    tm=$(echo "$out" | grep "Optimization time" | awk '{print $3" s"}')
    printf "%-12s %-4s %-24s %-18s %-8s %s\n" $f $a "$st" "$dobj" "$it" "$tm"
  done
done
rm -rf $T
