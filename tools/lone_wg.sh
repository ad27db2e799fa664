#!/bin/bash
# One workgroup per CU against two: what a lone workgroup's K loop makes of the matrix pipe (HDM_PERSIST_RESERVE_CUS=128 halves the
# persistent grid), with the operand traffic as it is and L2-resident (diagnostic build, HDM_VAR=192: wrong results, timing only).
R=${GRAFT_REPO_ROOT:-$(pwd)}
D=$R/hdsdp_amd/libhdsdp_mi355x_diag.so
for s in "" "HDM_PERSIST_RESERVE_CUS=128" "HDM_PERSIST_RESERVE_CUS=64" ; do
  echo "== product build [$s]"; env $s python3 $R/tools/build_timing.py 2000
done
if [ -f $D ]; then
  for s in "HDM_VAR=192" "HDM_VAR=192 HDM_PERSIST_RESERVE_CUS=128"; do
    echo "== diagnostic build [$s]"; env HDSDP_MI355X_LIB=$D $s python3 $R/tools/build_timing.py 2000
  done
fi
