#!/bin/bash
# kernel stats + the two SQ counter passes of a SMALL resident batch (development aid): tools/dev/prof_small.sh <tag> <reads>
set -u
TAG=${1:-small}; N=${2:-1024}
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS="--reads $N --steps 10 --warmup 2 --cpu-sample 0 --cpu-all-cores-sample 0 --no-overlap-test --no-api-leg --no-fast5-leg --no-e2e-leg --no-f32-leg --no-run-shaped-leg --no-full-leg --no-latency-leg"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python bench.py $ARGS > $OUT/trace.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $OUT/pmc1 -o p -- python bench.py $ARGS > $OUT/pmc1.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $OUT/pmc2 -o p -- python bench.py $ARGS > $OUT/pmc2.log 2>&1
python tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
python tools/prof_bounds.py $OUT/summary.txt k_scaler_lstm_q8_lat k_scaler_lstm_q8 k_demux_bidir_q8 k_demux_top_q8 k_viterbi_ltr > $OUT/bounds.json
cat $OUT/bounds.json
