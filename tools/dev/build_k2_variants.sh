#!/bin/bash
# A/B libraries of K2 (k_lstm_q8.hip) built with -DQ8S_SCHED=<v> (development aid): tools/dev/build_k2_variants.sh 0 2 3
# -> poreplex_amd/csrc/_exp/libpxg_s<v>.so, timed on the GPU box with tools/dev/ab.sh
set -e
C=/root/repo/poreplex_amd/csrc
[ -d "$C" ] || C="${GRAFT_REPO_ROOT}/poreplex_amd/csrc"
make -C $C -j8 --no-print-directory > /dev/null
mkdir -p $C/_exp
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall -Wno-unused-function -Wno-unused-variable"
for v in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS ${K2_EXTRA:-} -DQ8S_SCHED=$v -c $C/k_lstm_q8.hip -o $C/_exp/q8s_$v.o
  OBJS=$(ls $C/_obj/*.o | grep -v "/k_lstm_q8.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/_exp/libpxg_s$v.so $OBJS $C/_exp/q8s_$v.o
  echo built _exp/libpxg_s$v.so
done
