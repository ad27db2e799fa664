#!/bin/bash
for bc in ${BCS:-16 32 64}; do
  HDM_BC=$bc python bench.py --steps 3 --warmup 1 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; o=r['kernels']
print('BC=$bc', 'it/s', d['value'], 'ms', d['ms_per_step'], 'K1', o['congruence_step1']['ms_per_step'], 'K2', o['congruence_step2']['ms_per_step'], 'gram', o['gram']['ms_per_step'], 'frac', r['frac'])"
done
