#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for q in 0 96 128 160 200 224 252 288 336 400 504 672 1004 2004; do
  [ $q = 0 ] && unset PXG_SCALER_BLOCK_STEPS || export PXG_SCALER_BLOCK_STEPS=$q
  python bench.py --steps 8 --warmup 3 --cpu-sample 0 --cpu-all-cores-sample 0 --no-overlap-test --no-api-leg --no-fast5-leg --no-e2e-leg --no-f32-leg --no-run-shaped-leg ${SWEEP_ARGS:-} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['extra']['stage_ms']
print('QBS', '$q', 'K2', s['scaler_lstm'], 'step', round(d['ms_per_step'],3))"
done
