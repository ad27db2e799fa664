#!/bin/bash
# same-box A/B of environment switches: tools/ab.sh "HDM_NSPLIT=64" "HDM_NSPLIT=1024" "HDM_VAR=0" ...  (each run: bench.py --steps 3)
for cfg in "$@"; do
  env $cfg python bench.py --steps 3 --warmup 1 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; o=r['kernels']
print('$cfg', 'it/s', d['value'], 'ms', d['ms_per_step'], 'K1', o['congruence_step1']['ms_per_step'], 'K2', o['congruence_step2']['ms_per_step'], 'gram', o['gram']['ms_per_step'], 'frac', r['frac'])"
done
