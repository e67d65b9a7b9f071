#!/bin/bash
# a library built with extra compiler flags, for A/B timing (development aid):
#   tools/dev/build_variant.sh <name> <flags...>   -> poreplex_amd/csrc/_exp/libpxg_<name>.so   (tools/dev/ab.sh times it)
set -e
C="$(cd "$(dirname "$0")/../../poreplex_amd/csrc" && pwd)"
NAME=$1; shift
mkdir -p $C/_exp/$NAME
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall -Wno-unused-function -Wno-unused-variable"
for f in pxg_api k_signal k_viterbi k_lstm k_lstm_q8 k_lstm_q8_lat k_lstm_q8_demux k_polya k_unsplit; do
  EXTRA=""; [ $f = k_lstm_q8_demux ] && EXTRA="-mllvm -amdgpu-sched-strategy=iterative-ilp"
  /opt/rocm/bin/hipcc $FLAGS $EXTRA "$@" -c $C/$f.hip -o $C/_exp/$NAME/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/_exp/libpxg_$NAME.so $C/_exp/$NAME/*.o
echo built _exp/libpxg_$NAME.so
