#!/bin/bash
# Refresh of the round-6 profiles after the late K3 / K7b changes (development aid): rocprofv3 passes first, then the bench lines.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/final_r06b
mkdir -p $OUT
bash tools/prof.sh r06b_demux > /dev/null 2>&1
PXG_NO_SCAN_OVERLAP=1 PXG_NO_POLYA_OVERLAP=1 bash tools/prof.sh r06b_full_serial --workload full > /dev/null 2>&1
bash tools/prof.sh r06b_full --workload full > /dev/null 2>&1
B="python bench.py --cpu-sample 0 --cpu-all-cores-sample 0 --no-api-leg --no-fast5-leg --no-e2e-leg --no-f32-leg --no-run-shaped-leg --no-full-leg --no-latency-leg"
T0=$(date +%s)
python bench.py --steps 20 --warmup 5 > $OUT/bench_demux.json 2> $OUT/bench_demux.err
echo "driver form: $(( $(date +%s) - T0 )) s" > $OUT/wallclock.txt
for w in segment polya chimera full; do $B --workload $w --steps 10 --warmup 3 > $OUT/bench_$w.json 2> $OUT/bench_$w.err; done
$B --reads 100000 --steps 5 --warmup 2 > $OUT/bench_demux_100k_reads_60k_samples.json 2> $OUT/b100k.err
$B --workload full --reads 100000 --steps 5 --warmup 2 > $OUT/bench_full_100k_reads_60k_samples.json 2>> $OUT/b100k.err
$B --length-dist lognormal --steps 10 --warmup 3 > $OUT/bench_demux_lognormal.json 2> $OUT/lognormal.err
$B --length-dist lognormal --workload full --steps 10 --warmup 3 > $OUT/bench_full_lognormal.json 2>> $OUT/lognormal.err
$B --end-to-end --from-fast5 vbz --reads 120000 --batch-reads 10000 > $OUT/bench_end_to_end_fast5_vbz_12_batches.json 2> $OUT/e2e.err
for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1].split('/')[-1], round(d['value']), round(d['ms_per_step'], 3), (d.get('roofline') or {}).get('frac'))
except Exception as e:
    print(sys.argv[1], 'unreadable', e)
PY
done
