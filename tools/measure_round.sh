#!/bin/bash
# Every bench line + profile the round's profiles/ directory quotes, in one gpurun call.
# usage (on the GPU box): tools/measure_round.sh <tag>     -> gpurun_out/final_<tag>/
set -u
TAG=${1:-final}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/final_$TAG
mkdir -p $OUT
B="python bench.py --cpu-sample 0 --cpu-all-cores-sample 0 --no-api-leg --no-fast5-leg --no-e2e-leg --no-f32-leg --no-run-shaped-leg --no-full-leg --no-latency-leg"
T0=$(date +%s)
python bench.py --steps 20 --warmup 5 > $OUT/bench_demux.json 2> $OUT/bench_demux.err          # the driver's form
echo "driver form: $(( $(date +%s) - T0 )) s" > $OUT/wallclock.txt
for w in segment polya chimera full; do
  $B --workload $w --steps 10 --warmup 3 > $OUT/bench_$w.json 2> $OUT/bench_$w.err
done
$B --reads 100000 --steps 5 --warmup 2 > $OUT/bench_demux_100k_reads_60k_samples.json 2> $OUT/b100k.err
$B --workload full --reads 100000 --steps 5 --warmup 2 > $OUT/bench_full_100k_reads_60k_samples.json 2>> $OUT/b100k.err
$B --scaling strong --total-reads 125000 --steps 3 --warmup 1 > $OUT/bench_strong_125k_shard.json 2>> $OUT/b100k.err
$B --end-to-end --reads 120000 --batch-reads 10000 > $OUT/bench_end_to_end_demux.json 2> $OUT/e2e.err
$B --end-to-end --workload full --reads 120000 --batch-reads 10000 > $OUT/bench_end_to_end_full.json 2>> $OUT/e2e.err
$B --end-to-end --compressed-bundle --reads 120000 --batch-reads 10000 > $OUT/bench_end_to_end_demux_compressed.json 2>> $OUT/e2e.err
$B --end-to-end --compressed-bundle --workload full --reads 120000 --batch-reads 10000 > $OUT/bench_end_to_end_full_compressed.json 2>> $OUT/e2e.err
# round 3: the reference-shaped API, FAST5 input, the N > 1 path forced on one GPU
python bench.py --api process_batch --in-flight 5 --api-calls 24 --cpu-sample 0 --cpu-all-cores-sample 0 --no-overlap-test --no-full-leg --no-latency-leg --no-fast5-leg --no-e2e-leg > $OUT/bench_api_process_batch.json 2> $OUT/api.err
python bench.py --api process_batch --workload full --in-flight 5 --api-calls 12 --cpu-sample 0 --cpu-all-cores-sample 0 --no-overlap-test --no-full-leg --no-latency-leg --no-fast5-leg --no-e2e-leg > $OUT/bench_api_process_batch_full.json 2>> $OUT/api.err
for c in none vbz gzip; do
  r=120000; [ $c = vbz ] && r=60000; [ $c = gzip ] && r=30000        # (the pure-Python FAST5 writer is what takes the time here)
  $B --end-to-end --from-fast5 $c --reads $r --batch-reads 10000 > $OUT/bench_end_to_end_fast5_$c.json 2>> $OUT/e2e.err
done
tools/dev/e2e_fast5.sh $OUT/e2e_fast5_per_batch.txt none 300000 1 > /dev/null 2>&1      # 30 batches, the loader's time per batch
python tools/fast5_ingest_profile.py 10000 > $OUT/fast5_ingest_profile.txt 2>&1
# round 4: the float32 arithmetic, run-shaped lengths, the bare --gpus 8 command under the driver's clock
$B --lstm-arith f32 --steps 10 --warmup 3 > $OUT/bench_demux_f32_arith.json 2> $OUT/f32.err
$B --length-dist lognormal --steps 10 --warmup 3 > $OUT/bench_demux_lognormal.json 2> $OUT/lognormal.err
$B --length-dist lognormal --workload full --steps 10 --warmup 3 > $OUT/bench_full_lognormal.json 2>> $OUT/lognormal.err
python bench.py --api process_batch --length-dist lognormal --in-flight 5 --api-calls 24 --cpu-sample 0 --cpu-all-cores-sample 0 --no-overlap-test --no-full-leg --no-latency-leg --no-fast5-leg --no-e2e-leg --no-f32-leg --no-run-shaped-leg > $OUT/bench_api_process_batch_lognormal.json 2>> $OUT/api.err
T1=$(date +%s)
PXG_BENCH_SHARE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 8 --steps 20 --warmup 5 > $OUT/bench_8_ranks_sharing_one_gpu.json 2> $OUT/share8.err
echo "bare --gpus 8 (--steps 20 --warmup 5) with 8 ranks on ONE GPU: $(( $(date +%s) - T1 )) s" >> $OUT/wallclock.txt
PXG_BENCH_FORCE_DIST=1 python bench.py --steps 10 --warmup 3 --cpu-sample 128 --cpu-all-cores-sample 0 --no-api-leg --no-fast5-leg --no-e2e-leg --no-full-leg --no-latency-leg > $OUT/bench_force_dist_one_gpu.json 2> $OUT/dist.err
# round 6: small batches (latency forms of K2 / K5) and the reference-sized API with and without them
python tools/dev/small_batch.py 128 512 1024 2048 4096 > $OUT/small_batch_stage_ms.txt 2>&1
(tools/dev/api128.sh PXG_K2_LAT_MAX=0 PXG_K5_LAT_MAX=0; tools/dev/api128.sh PXG_X=1) > $OUT/api_128_read_calls.txt 2>&1
if [ -z "${SKIP_PROF:-}" ]; then
  bash tools/prof.sh ${TAG}_demux > /dev/null 2>&1
  PXG_NO_SCAN_OVERLAP=1 PXG_NO_POLYA_OVERLAP=1 bash tools/prof.sh ${TAG}_full_serial --workload full > /dev/null 2>&1
  bash tools/prof.sh ${TAG}_full --workload full > /dev/null 2>&1
  tools/dev/prof_small.sh ${TAG}_small1024 1024 > /dev/null 2>&1
fi
for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1].split('/')[-1], round(d['value']), round(d['ms_per_step'], 3), (d.get('roofline') or {}).get('frac'))
except Exception as e:
    print(sys.argv[1], 'unreadable', e)
PY
done
