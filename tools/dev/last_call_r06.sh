#!/bin/bash
# One gpurun call for the tree as it stands at the end of round 6 (written while GPU access was closed): the whole GPU
# suite, smoke, the driver's form of bench.py, the reference-sized API with and without the plain-run path, and the VBZ
# session over 24 batches.   usage (on the GPU box): tools/dev/last_call_r06.sh   -> gpurun_out/r6_last/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r6_last
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1
tail -3 $OUT/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1
tail -1 $OUT/smoke.txt
T0=$(date +%s)
timeout 900 python bench.py > $OUT/bench_demux.json 2> $OUT/bench_demux.err
echo "driver form: $(( $(date +%s) - T0 )) s" | tee $OUT/wallclock.txt
(API_FL="1 32" tools/dev/api128.sh PXG_NO_PLAIN_RUN=1; API_FL="1 32" tools/dev/api128.sh PXG_X=1) > $OUT/api_128_read_calls_plain_run.txt 2>&1
cat $OUT/api_128_read_calls_plain_run.txt
# 128-read worker calls over multi-read FAST5 files (no bundle), real context: batch table / plain run / + fused decode and pass (shipped) / + page-locked call arenas
(for env in PXG_NO_PLAIN_RUN_FAST5=1 PXG_NO_FUSED_CALL=1 PXG_X=1 PXG_PIN_CALL_ARENAS=1; do echo "## $env"; env $env timeout 600 python tools/dev/host_cap.py --fast5 none --real --file-reads 4096 2>&1 | tail -4; done
 echo "## vbz"; timeout 600 python tools/dev/host_cap.py --fast5 vbz --real --file-reads 4096 2>&1 | tail -4) > $OUT/api_128_read_calls_fast5.txt 2>&1
cat $OUT/api_128_read_calls_fast5.txt
B="python bench.py --cpu-sample 0 --cpu-all-cores-sample 0 --no-api-leg --no-fast5-leg --no-e2e-leg --no-f32-leg --no-run-shaped-leg --no-full-leg --no-latency-leg"
timeout 900 $B --end-to-end --from-fast5 vbz --reads 240000 --batch-reads 10000 > $OUT/bench_end_to_end_fast5_vbz_24_batches.json 2> $OUT/e2e.err
python - $OUT <<'PY'
import json, sys
out = sys.argv[1]
for name in ('bench_demux.json', 'bench_end_to_end_fast5_vbz_24_batches.json'):
    try:
        d = json.loads(open(out + '/' + name).read().strip().splitlines()[-1])
        print(name, round(d['value']), round(d['ms_per_step'], 3), (d.get('roofline') or {}).get('frac'))
        lf = (d.get('roofline') or {}).get('latency_form', {}).get('process_batch_128_read_calls')
        if lf:
            print('  128-read calls:', json.dumps(lf))
    except Exception as e:
        print(name, 'unreadable', e)
PY
