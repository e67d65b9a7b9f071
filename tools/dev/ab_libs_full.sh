#!/bin/bash
# A/B of library builds on the chimera and full workloads (development aid): tools/dev/ab_libs_full.sh <lib> ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2; do for lib in "$@"; do for wl in chimera full; do
  PXG_LIBRARY=$PWD/$lib python bench.py --workload $wl --steps 10 --warmup 3 ${AB_ARGS:-} --cpu-sample ${AB_CPU_SAMPLE:-32} --cpu-all-cores-sample 0 --no-overlap-test --no-api-leg --no-fast5-leg --no-e2e-leg --no-f32-leg --no-run-shaped-leg --no-full-leg --no-latency-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
s=d['extra']['stage_ms']
print('$lib $wl', round(d['value']), 'ms/step', round(d['ms_per_step'],3), {k: s[k] for k in ('polya','event_means','unsplit','total')}, 'bit-exact', (d.get('concordance') or {}).get('all_fields_bit_exact'), (d.get('concordance') or {}).get('unsplit_candidate_mismatch'))"
done; done; done
