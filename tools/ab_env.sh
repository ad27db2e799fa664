#!/bin/bash
# same box A/B of bench.py under environment settings: tools/ab_env.sh "VAR=1" "VAR=0" ...  (each argument: one setting or several
# separated by blanks, or "" for none; the first is repeated at the end).  BENCH_ARGS="--m 8000" adds bench.py arguments,
# AB_STEPS the timed steps (default 4).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/ab_env
mkdir -p $O
i=0
for setting in "$@" "$1"; do
  i=$((i+1))
  env $setting python3 $R/bench.py --steps ${AB_STEPS:-4} --warmup 1 --no-cpu $BENCH_ARGS > $O/run$i.json 2> $O/run$i.err
  python3 - <<PY
import json
try:
    d=json.loads(open("$O/run$i.json").read().strip().splitlines()[-1])
    pc=d["roofline"].get("power_clock") or {}
    print("[$setting]", d["ms_per_step"], {k:(v["ms_per_step"], v["tflops"]) for k,v in d["roofline"]["kernels"].items()},
          "W", [round(x) for x in pc.get("board_power_w",[])], "MHz", [round(x) for x in pc.get("shader_clock_mhz",[])],
          d["checksum"]["sum_d2"], d.get("checksum_ok"), flush=True)
except Exception as e:
    print("[$setting] FAILED", e, open("$O/run$i.err").read()[-400:], flush=True)
PY
done
