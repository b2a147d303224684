#!/bin/bash
# quick look at the lone-caller host path (pinned and pageable) + the respond tests
set -e
cd "$(dirname "$0")/.."
timeout -k 10 600 python -m pytest tests/test_gpu_respond.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -n 5
for pinned in 1 0; do
  CPIR_RESPOND_TRACE=1 timeout -k 10 120 python3 scripts/host_path_probe.py ${1:-cfg2} 1 400 $pinned 1 2>&1 | grep -v amdgpu.ids | cut -c1-200
done
