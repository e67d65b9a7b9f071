#!/bin/bash
# A/B timing of library builds on the poly(A) workload (development aid): tools/dev/ab_polya.sh <lib> ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for lib in "$@"; do
  PXG_LIBRARY=$PWD/$lib python bench.py --workload ${AB_WORKLOAD:-polya} --steps 8 --warmup 3 --cpu-sample 48 --cpu-all-cores-sample 0 --no-overlap-test --no-api-leg --no-fast5-leg --no-e2e-leg --no-f32-leg --no-run-shaped-leg --no-full-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['extra']['stage_ms']; c=d.get('concordance') or {}
print('$lib', round(d['value']), 'ms/step', round(d['ms_per_step'],3), 'K6', s['polya'], 'K7a', s['event_means'], 'K7b', s['unsplit'], 'bit-exact', c.get('all_fields_bit_exact'), c.get('unsplit_candidate_mismatch'))"
done
