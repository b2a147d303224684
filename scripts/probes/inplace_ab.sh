#!/bin/bash
# A/B of the in-place rounds (respond.inplace_seats 0 vs 4): T closed-loop callers with page-locked / pageable queries through
# examples/host_respond_bench.c.   usage: scripts/probes/inplace_ab.sh [lg_keys [value_bytes [callers...]]]   (CPIR_BENCH_HOLES=9: compacted server)
LG=${1:-20}; VB=${2:-1024}; shift 2 2>/dev/null
CALLERS=${@:-2 3 4 8}
mkdir -p gpurun_out/r5
out=gpurun_out/r5/inplace_ab_${LG}_${VB}.txt
: > $out
for T in $CALLERS; do
  for seats in 0 4; do
    for pinned in 1 0; do
      echo -n "2^$LG keys x $VB B, callers $T inplace_seats $seats pinned $pinned: " >> $out
      CPIR_BENCH_INPLACE_SEATS=$seats timeout -k 10 300 chalametpir_amd/lib/host_respond_bench $LG $VB 3 t$T $pinned >> $out 2>&1 || exit 1
    done
  done
done
cat $out
