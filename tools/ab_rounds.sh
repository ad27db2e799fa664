#!/bin/bash
# same-box comparison of the previous round's final tree (_r04/, commit 9f2fc68 unpacked and built beside this one: its own bench.py,
# library and solver harness) and this tree: the bench at both BASELINE sizes and the headline solve by the reference's unchanged driver
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/ab_rounds
mkdir -p $O
show() {
python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = d["roofline"]["kernels"]
    w = d["roofline"].get("congruence_step2_whole", {})
    pc = d["roofline"].get("power_clock") or {}
    print("%-22s %9.3f ms/step %7.4f it/s  K1 %7.2f  K2 %7.2f  gram %8.2f (%.1f TF)  clock %s  checksum_ok %s" % (sys.argv[2], d["ms_per_step"], d["value"],
          k["congruence_step1"]["ms_per_step"], w.get("ms_per_step", k["congruence_step2"]["ms_per_step"]), k["gram"]["ms_per_step"], k["gram"]["tflops"],
          [round(x) for x in pc.get("shader_clock_mhz", [])], d.get("checksum_ok")), flush=True)
except Exception as e:
    print(sys.argv[2], "no result:", e, flush=True)
PY
}
for rep in 1 2; do
  for which in r04 r05; do
    d=$R; [ $which = r04 ] && d=$R/_r04
    (cd $d && python3 bench.py --steps 6 --warmup 2 --no-cpu > $O/${which}_2000_$rep.json 2> $O/${which}_2000_$rep.err) || echo "$which $rep failed"
    show $O/${which}_2000_$rep.json "$which n=m=2000 run $rep"
  done
done
for which in r04 r05 r04 r05; do
  d=$R; [ $which = r04 ] && d=$R/_r04
  (cd $d && python3 bench.py --m 8000 --steps 3 --warmup 1 --no-cpu > $O/${which}_8000.json 2> $O/${which}_8000.err) || echo "$which 8000 failed"
  show $O/${which}_8000.json "$which m=8000"
done
for which in r04 r05; do
  d=$R; [ $which = r04 ] && d=$R/_r04
  (cd $d && HDSDP_DROP_ATTACH=1 HDSDP_MI355X_CALL_STATS=1 oracle/_ref/headline_solve_mi355x 2000 > $O/${which}_solve.log 2>&1)
  echo "$which headline solve (attach = 1): $(grep -E 'Pre-solver ends|Optimization time|wall time below' $O/${which}_solve.log | tr '\n' ' ')"
done
(cd $R && HDSDP_DROP_ATTACH=2 HDSDP_MI355X_CALL_STATS=1 oracle/_ref/headline_solve_mi355x 2000 > $O/r05_solve2.log 2>&1)
echo "r05 headline solve (attach = 2): $(grep -E 'Pre-solver ends|Optimization time|wall time below' $O/r05_solve2.log | tr '\n' ' ')"
