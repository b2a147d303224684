#!/bin/bash
# host path: lone and concurrent callers, pageable and pinned, query read in place on and off
set -e
cd "$(dirname "$0")/.."
cfg=${1:-cfg2}
for zc in 1 0; do
  for pinned in 1 0; do
    for th in 1 2 8; do
      CPIR_RESPOND_TRACE=1 timeout -k 10 120 python3 scripts/host_path_probe.py $cfg $th 400 $pinned $zc 2>&1 | cut -c1-600
    done
  done
done
