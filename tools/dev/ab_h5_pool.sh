#!/bin/bash
# A/B of the FAST5 reader's host threads (development aid, host cores of a GPU box): shared worker
# threads (default) against threads per call (PXG_H5_SPAWN_THREADS=1) -- the ingest profile by phase,
# then the session from uncompressed multi-read FAST5 over 12 batches.   tools/dev/ab_h5_pool.sh <out dir>
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=${1:-gpurun_out/ab_h5_pool}; mkdir -p $OUT
for mode in pool spawn; do
  if [ $mode = spawn ]; then export PXG_H5_SPAWN_THREADS=1; else unset PXG_H5_SPAWN_THREADS; fi
  echo "== $mode" >> $OUT/ingest_profile.txt
  PXG_PROF_MODES=${AB_MODES:-none,vbz} PXG_H5_TRACE=1 python tools/fast5_ingest_profile.py 10000 >> $OUT/ingest_profile.txt 2>&1
  for rep in 1 2; do
    python bench.py --end-to-end --from-fast5 none --reads 120000 --batch-reads 10000 --cpu-sample 0 --cpu-all-cores-sample 0 2>> $OUT/e2e.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode', round(d['value']), d['extra']['session_timing_rank0'])" >> $OUT/e2e.txt
  done
done
cat $OUT/ingest_profile.txt | grep -v "^pxg_h5_open" ; grep "pxg_h5_open" $OUT/ingest_profile.txt | sort | uniq -c | sort -rn | head -12; cat $OUT/e2e.txt
