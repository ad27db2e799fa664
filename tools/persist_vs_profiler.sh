#!/bin/bash
# same box: bench.py plain, under rocprofv3 --kernel-trace --stats, plain again (does the profiler change the persistent kernels?)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pvp
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 4 --warmup 1 --no-cpu > $O/plain1.json 2> $O/plain1.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu > $O/prof.json 2> $O/prof.err
python3 $R/bench.py --steps 4 --warmup 1 --no-cpu > $O/plain2.json 2> $O/plain2.err
HDM_PERSIST=0 python3 $R/bench.py --steps 4 --warmup 1 --no-cpu > $O/oneshot.json 2> $O/oneshot.err
rm -rf $O/stats
python3 - <<PY
import json
for f in ("plain1","prof","plain2","oneshot"):
    d=json.loads(open("$O/%s.json"%f).read().strip().splitlines()[-1])
    print(f, d["ms_per_step"], {k:v["ms_per_step"] for k,v in d["roofline"]["kernels"].items()})
PY
