#!/bin/bash
# Copies what scripts/profile_gpu.sh left under gpurun_out/prof_<tag>{,_cfg5} (and the bench lines gpurun_out/<tag>_bench_*.json) into
# profiles/ under the names profiles/README.md lists, and merges the traffic records.   usage: scripts/collect_profiles.sh <tag>
set -e
cd "$(dirname "$0")/.."
T=$1
for v in "" _cfg3 _cfg4 _cfg5; do
  P=gpurun_out/prof_$T$v
  [ -d "$P" ] || continue
  cp $P/trace/trace_kernel_stats.csv profiles/${T}${v}_kernel_stats.csv
  [ -f $P/setup_trace/setup_kernel_stats.csv ] && cp $P/setup_trace/setup_kernel_stats.csv profiles/${T}${v}_setup_kernel_stats.csv
  [ -f $P/full_trace/full_kernel_stats.csv ] && cp $P/full_trace/full_kernel_stats.csv profiles/${T}${v}_all_sections_kernel_stats.csv
  cp $P/summary.txt profiles/${T}${v}_rocprofv3_summary.txt
  cp $P/summary.json profiles/${T}${v}_rocprofv3_summary.json
  cp $P/trace_bench.json profiles/${T}${v}_bench_under_kernel_trace.json
  python3 scripts/merge_traffic.py $P/respond_traffic.json
done
for f in gpurun_out/${T}_bench_*.json; do [ -s "$f" ] && cp "$f" profiles/; done
ls profiles | grep "^$T"
