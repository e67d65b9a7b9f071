#!/bin/bash
# A/B of library builds on the segment workload + default line (development aid): tools/dev/ab_segment.sh <lib> ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2; do for lib in "$@"; do
  for wl in segment demux; do
  PXG_LIBRARY=$PWD/$lib python bench.py --workload $wl --steps 10 --warmup 3 --cpu-sample ${AB_CPU_SAMPLE:-32} --cpu-all-cores-sample 0 --no-overlap-test --no-api-leg --no-fast5-leg --no-e2e-leg --no-f32-leg --no-run-shaped-leg --no-full-leg --no-latency-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
s=d['extra']['stage_ms']
print('$lib $wl', round(d['value']), 'ms/step', round(d['ms_per_step'],3), 'K3', s['segment'], 'K2', s['scaler_lstm'], 'bit-exact', (d.get('concordance') or {}).get('all_fields_bit_exact'))"
  done
done; done
