#!/bin/bash
# A/B timing of library builds (development aid): tools/dev/ab.sh <lib> [<lib> ...]  (paths relative to the repo)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for lib in "$@"; do
  PXG_LIBRARY=$PWD/$lib python bench.py --steps 10 --warmup 3 --cpu-sample ${AB_CPU_SAMPLE:-32} --cpu-all-cores-sample 0 --no-overlap-test --no-api-leg --no-fast5-leg --no-e2e-leg --no-f32-leg ${AB_ARGS:-} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
s=d['extra']['stage_ms']
print('$lib', round(d['value']), 'ms/step', round(d['ms_per_step'],3), 'K2', s['scaler_lstm'], 'K5a', s['demux_bidir'], 'K5b', s['demux_top'], 'K3', s['segment'], 'bit-exact', (d.get('concordance') or {}).get('all_fields_bit_exact'))"
done
