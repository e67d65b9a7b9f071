#!/bin/bash
# process_batch at small reads-per-call (the reference's default --batch-size is 128), many calls in flight,
# with and without the merging of small calls: usage tools/dev/api_small_calls.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for n in 128 512; do for fl in 8 32 64; do for nm in 0 1; do
  if [ $nm = 1 ]; then export PXG_NO_CALL_MERGE=1; else unset PXG_NO_CALL_MERGE; fi
  timeout 300 python bench.py --api process_batch --reads $n --in-flight $fl --api-calls 640 --cpu-sample 0 --cpu-all-cores-sample 0 \
    --no-overlap-test --no-fast5-leg --no-e2e-leg --no-f32-leg --no-run-shaped-leg 2>/tmp/api_err.txt > /tmp/api_$n.json || tail -3 /tmp/api_err.txt
  python - $n $fl $nm <<'PY'
import json, sys
n, fl, nm = sys.argv[1:4]
d = json.loads(open('/tmp/api_%s.json' % n).read().strip().splitlines()[-1]); e = d['extra']
print(n, 'reads per call,', fl, 'in flight, merge', 'off' if nm == '1' else 'on', ': process_batch raw', round(e['process_batch_reads_per_s']),
      'encoded', round(e['process_batch_encoded_bundle_reads_per_s']), json.dumps(e['process_batch']['raw'].get('mean_phase_ms_per_call')),
      e['process_batch']['raw'].get('merge_stats'))
PY
done; done; done
