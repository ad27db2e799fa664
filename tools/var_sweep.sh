#!/bin/bash
# A/B kernel variants on the GPU box: HDM_VAR bit0 = s_setprio around the MFMA block, bit1 = early LDS write
for v in ${VARS:-0 1 2 3}; do
  HDM_VAR=$v python bench.py --steps 3 --warmup 1 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; o=r['kernels']
print('VAR=$v', 'it/s', d['value'], 'ms', d['ms_per_step'], 'K1', o['congruence_step1']['ms_per_step'], 'K2', o['congruence_step2']['ms_per_step'], 'gram', o['gram']['ms_per_step'], 'dom', r['kernel'][:34], 'frac', r['frac'])"
done
