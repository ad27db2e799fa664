#!/bin/bash
# same-box comparison of the round-2 final tree (_r02/, commit e9dc045, its own bench.py and library) and this tree
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/ab_rounds
mkdir -p $O
for rep in 1 2; do
  for which in r02 r03; do
    d=$R; [ $which = r02 ] && d=$R/_r02
    (cd $d && python3 bench.py --steps 6 --warmup 2 --no-cpu > $O/${which}_$rep.json 2> $O/${which}_$rep.err) || echo "$which $rep failed"
    python3 - "$O/${which}_$rep.json" "$which run $rep" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = d["roofline"]["kernels"]
    w = d["roofline"].get("congruence_step2_whole", {})
    print("%-12s %.3f ms/step  %.3f it/s  K1 %.2f  K2 %.2f  gram %.2f  stages %s  sum_d2 %.16g" % (sys.argv[2], d["ms_per_step"], d["value"], k["congruence_step1"]["ms_per_step"],
          w.get("ms_per_step", k["congruence_step2"]["ms_per_step"]), k["gram"]["ms_per_step"], d["config"]["stage_ms"], d["checksum"]["sum_d2"]))
except Exception as e:
    print(sys.argv[2], "no result:", e)
PY
  done
done
