#!/bin/bash
# average outstanding time of an L2 read towards the fabric (TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ) of every GEMM role: bench.py [args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/lat_all -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu "$@" > $R/gpurun_out/lat_all.json 2> $R/gpurun_out/lat_all.err
python3 $R/tools/prof_summary.py $R/gpurun_out/lat_all "ea latency" 2>&1 | grep -E "persist_kernel<.*GRBM" | python3 -c "
import sys, re
for l in sys.stdin:
    name = l[:64].strip()
    lv = float(re.search(r'TCC_EA0_RDREQ_LEVEL_sum=([0-9.e+]+)', l).group(1)); rq = float(re.search(r'TCC_EA0_RDREQ_sum=([0-9.e+]+)', l).group(1))
    print('%-66s reads per launch %.3e  cycles per read %.0f' % (name, rq, lv / rq))
"
rm -rf $R/gpurun_out/lat_all
