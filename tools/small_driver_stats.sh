#!/bin/bash
# time below the C ABI of the reference's unchanged driver on the small instances (engine cones attached), by entry point:
# tools/small_driver_stats.sh [ENV=VAL ...]    (HDSDP_MI355X_CALL_STATS=1 prints the table at exit)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for kv in "$@"; do export "$kv"; done
for f in theta1 mcp100 gpp100 truss1; do
  echo "== $f $*"
  HDSDP_DROP_ATTACH=1 HDSDP_MI355X_CALL_STATS=1 timeout -k 10 300 $R/oracle/_ref/sdpasolve_mi355x $R/tests/golden/$f.dat-s 2>&1 | grep -E "hdsdp_mi355x\]|SDP Status|dObj|Optimization time" | grep -v "device group"
done
