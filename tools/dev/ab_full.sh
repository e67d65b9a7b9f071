#!/bin/bash
# A/B of environment settings on the `full` workload (development aid): tools/dev/ab_full.sh "ENV=1" "OTHER=1" ...  ("-" = no setting)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2; do for e in "$@"; do
  [ "$e" = "-" ] && E="PXG_X=1" || E="$e"
  env $E python bench.py --workload full --steps 10 --warmup 3 ${AB_ARGS:-} --cpu-sample ${AB_CPU_SAMPLE:-32} --cpu-all-cores-sample 0 --no-overlap-test --no-api-leg --no-fast5-leg --no-e2e-leg --no-f32-leg --no-run-shaped-leg --no-full-leg --no-latency-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
s=d['extra']['stage_ms']
print('$e', round(d['value']), 'ms/step', round(d['ms_per_step'],3), {k: s[k] for k in ('scaler_lstm','segment','demux_bidir','demux_top','polya','event_means','unsplit','finalize','total')}, 'bit-exact', (d.get('concordance') or {}).get('all_fields_bit_exact'))"
done; done
