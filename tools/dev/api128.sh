#!/bin/bash
# process_batch at the reference's 128 reads per call, 1 / 32 / 64 calls in flight (development aid): tools/dev/api128.sh [env assignments...]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for fl in ${API_FL:-1 32 64}; do
  timeout 300 env "$@" python bench.py --api process_batch --reads 128 --in-flight $fl --api-calls ${API_CALLS:-640} --cpu-sample 0 --cpu-all-cores-sample 0 \
    --no-overlap-test --no-fast5-leg --no-e2e-leg --no-f32-leg --no-run-shaped-leg --no-full-leg 2>/tmp/api_err.txt > /tmp/api_128.json || tail -3 /tmp/api_err.txt
  python - $fl "$*" <<'PY'
import json, sys
fl, env = sys.argv[1:3]
d = json.loads(open('/tmp/api_128.json').read().strip().splitlines()[-1]); e = d['extra']
r = e['process_batch']['raw']
print(env, '| 128 reads per call,', fl, 'in flight: raw', round(e['process_batch_reads_per_s']), 'one-at-a-time', round(r['one_call_at_a_time_reads_per_s']),
      'encoded', round(e['process_batch_encoded_bundle_reads_per_s']), json.dumps(r.get('mean_phase_ms_per_call')), r.get('merge_stats'), flush=True)
PY
done
