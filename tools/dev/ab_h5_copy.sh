#!/bin/bash
# A/B of the FAST5 reader's bulk copy (development aid, host cores of a GPU box): PXG_H5_COPY modes through the
# ingest profile and the session from uncompressed FAST5.   tools/dev/ab_h5_copy.sh <out dir> [modes...]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=${1:-gpurun_out/ab_h5_copy}; shift; mkdir -p $OUT
for mode in ${@:-pread bounce}; do
  export PXG_H5_COPY=$mode
  echo "== $mode" >> $OUT/ingest_profile.txt
  PXG_PROF_MODES=${AB_FAST5:-none} python tools/fast5_ingest_profile.py 10000 >> $OUT/ingest_profile.txt 2>&1
  for rep in 1 2; do
    python bench.py --end-to-end --from-fast5 ${AB_FAST5:-none} --reads ${AB_READS:-120000} --batch-reads 10000 --cpu-sample 0 --cpu-all-cores-sample 0 2>> $OUT/e2e.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode', round(d['value']), d['extra']['session_timing_rank0'])" >> $OUT/e2e.txt
  done
done
cat $OUT/ingest_profile.txt $OUT/e2e.txt
