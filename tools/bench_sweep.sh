#!/bin/bash
# usage: sweep.sh <workload> <stage> lib1 lib2 ...
wl=$1; st=$2; shift 2
for lib in "$@"; do
  PXG_LIBRARY=$lib timeout 300 python bench.py --workload $wl --steps 5 --cpu-sample 0 2>&1 | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', round(d['value']), {k: d['extra']['stage_ms'][k] for k in '$st'.split(',')})"
done
