#!/bin/bash
# lone caller, pinned buffer: query read in place against upload first, at several configs
cd "$(dirname "$0")/.."
for cfg in "$@"; do
  for zc in 1 0; do
    CPIR_RESPOND_TRACE=1 timeout -k 10 200 python3 scripts/host_path_probe.py $cfg 1 200 1 $zc 2>&1 | grep -v amdgpu.ids | cut -c1-330
  done
done
