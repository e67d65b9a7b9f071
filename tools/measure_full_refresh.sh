#!/bin/bash
# The lines and profiles of the configs[3] stages only (a late change to K7b / the scan stream): tools/measure_full_refresh.sh <tag>
set -u
TAG=${1:-refresh}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/final_$TAG
mkdir -p $OUT
B="python bench.py --cpu-sample 0 --cpu-all-cores-sample 0 --no-api-leg --no-fast5-leg --no-e2e-leg --no-f32-leg --no-run-shaped-leg --no-full-leg --no-latency-leg"
python bench.py --steps 20 --warmup 5 > $OUT/bench_demux.json 2> $OUT/bench_demux.err
for w in polya chimera full; do $B --workload $w --steps 10 --warmup 3 > $OUT/bench_$w.json 2> $OUT/bench_$w.err; done
PXG_NO_SCAN_OVERLAP=1 $B --workload full --steps 10 --warmup 3 > $OUT/bench_full_scan_behind_run.json 2>> $OUT/bench_full.err
$B --workload full --reads 100000 --steps 5 --warmup 2 > $OUT/bench_full_100k_reads_60k_samples.json 2>> $OUT/bench_full.err
$B --length-dist lognormal --workload full --steps 10 --warmup 3 > $OUT/bench_full_lognormal.json 2>> $OUT/bench_full.err
$B --end-to-end --workload full --reads 120000 --batch-reads 10000 > $OUT/bench_end_to_end_full.json 2> $OUT/e2e.err
$B --end-to-end --compressed-bundle --workload full --reads 120000 --batch-reads 10000 > $OUT/bench_end_to_end_full_compressed.json 2>> $OUT/e2e.err
python bench.py --api process_batch --workload full --in-flight 5 --api-calls 12 --cpu-sample 0 --cpu-all-cores-sample 0 --no-overlap-test --no-full-leg --no-latency-leg --no-fast5-leg --no-e2e-leg > $OUT/bench_api_process_batch_full.json 2> $OUT/api.err
PXG_NO_SCAN_OVERLAP=1 bash tools/prof.sh ${TAG}_full_serial --workload full > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/prof_${TAG}_full_overlap
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_full_overlap/trace -o t -- $B --workload full --steps 10 --warmup 2 --no-overlap-test > gpurun_out/prof_${TAG}_full_overlap/trace.log 2>&1
for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1].split('/')[-1], round(d['value']), round(d['ms_per_step'], 3), (d.get('roofline') or {}).get('frac'))
except Exception as e:
    print(sys.argv[1], 'unreadable', e)
PY
done
