#!/bin/bash
# same box A/B of bench.py under environment settings: tools/ab_env.sh "VAR=1" "VAR=0" ...  (each argument: one setting, or "" for none)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/ab_env
mkdir -p $O
i=0
for setting in "$@" "$1"; do
  i=$((i+1))
  env $setting python3 $R/bench.py --steps 4 --warmup 1 --no-cpu > $O/run$i.json 2> $O/run$i.err
  python3 - <<PY
import json
d=json.loads(open("$O/run$i.json").read().strip().splitlines()[-1])
print("[$setting]", d["ms_per_step"], {k:v["ms_per_step"] for k,v in d["roofline"]["kernels"].items()}, d["checksum"]["sum_d2"])
PY
done
