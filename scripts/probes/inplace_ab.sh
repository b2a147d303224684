#!/bin/bash
# A/B of the in-place rounds: T closed-loop callers with page-locked / pageable queries, respond.inplace_seats 0 vs 4
mkdir -p gpurun_out/r5
out=gpurun_out/r5/inplace_ab.txt
: > $out
for T in 2 3 4 8; do
  for seats in 0 4; do
    for pinned in 1 0; do
      echo -n "callers $T inplace_seats $seats pinned $pinned: " >> $out
      CPIR_BENCH_INPLACE_SEATS=$seats timeout -k 10 120 chalametpir_amd/lib/host_respond_bench 20 1024 3 t$T $pinned >> $out 2>&1 || exit 1
    done
  done
done
cat $out
