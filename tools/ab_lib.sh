#!/bin/bash
# same-box A/B of two builds of the library: tools/ab_lib.sh TAG libA.so libB.so [steps]   (alternating runs, bench.py --no-cpu)
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; A=$2; B=$3; steps=${4:-6}
O=$R/gpurun_out/ab_$tag
mkdir -p $O
for rep in 1 2; do
  for which in A B; do
    lib=$A; [ $which = B ] && lib=$B
    HDSDP_MI355X_LIB=$R/$lib python3 $R/bench.py --steps $steps --warmup 2 --no-cpu > $O/${which}_$rep.json 2> $O/${which}_$rep.err || echo "$which $rep failed"
    python3 - "$O/${which}_$rep.json" "$which$rep $lib" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = d["roofline"]["kernels"]
    w = d["roofline"].get("congruence_step2_whole", {})
    print("%-40s %.3f ms/step  K1 %.2f  K2 %.2f (diagonal tiles %.2f)  gram %.2f  checksum_ok %s  sum_d2 %.16g" % (sys.argv[2], d["ms_per_step"], k["congruence_step1"]["ms_per_step"],
          w.get("ms_per_step", k["congruence_step2"]["ms_per_step"]), w.get("diagonal_tiles_ms_per_step", 0.0), k["gram"]["ms_per_step"], d.get("checksum_ok"), d["checksum"]["sum_d2"]))
except Exception as e:
    print(sys.argv[2], "no result:", e)
PY
  done
done
