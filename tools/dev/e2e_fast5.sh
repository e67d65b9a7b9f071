#!/bin/bash
# The session from multi-read FAST5 files, whole runs with the loader's per-batch times (development aid):
# tools/dev/e2e_fast5.sh <out file> <compression: none|vbz|gzip> <reads> [repeats]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$1; mkdir -p $(dirname $OUT)
for rep in $(seq 1 ${4:-2}); do
  python bench.py --end-to-end --from-fast5 $2 --reads $3 --batch-reads 10000 --cpu-sample 0 --cpu-all-cores-sample 0 2>> $OUT.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['extra']['session_timing_rank0']; ph=t.pop('load_phases_ms')
steady=sorted(t['load_ms'][2:]); med=steady[len(steady)//2] if steady else None
print('$2', '$3 reads:', round(d['value']), 'reads/s; loader ms per batch (median from the third on)', med, '; close', d['extra']['session_close_s'], 's;', t)
print('   per batch [walk (prefetch thread), signals, text, wait for prefetch, prepare]:', ph[:4], '...', ph[-2:])" >> $OUT
done
cat $OUT
