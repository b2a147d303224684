#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel trace + HBM traffic counters for the bench's respond kernel.
# Counters are collected in their own passes (never combined with trace domains other than kernel-trace), as the pool requires.
# Usage: scripts/profile_gpu.sh <tag> <git head (the box has no .git)> [extra bench args, e.g. --config cfg5]
# Writes gpurun_out/prof_<tag>/{summary.txt,summary.json,respond_traffic.json,...}; copy what is to be judged into profiles/ and merge
# respond_traffic.json into profiles/respond_traffic.json with scripts/merge_traffic.py.
set -u
TAG=${1:-r01}; shift || true
HEAD=${1:-unknown}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# the HEADLINE's timed loop and nothing else: since round 5 every device-resident launch runs on one kernel (respond_planar_wide_kernel), so a
# trace that also held the fused batches and the lone launches would average three different launch shapes under one name; here every
# dispatch of the kernel is one step of the timed loop (32 passes), and its average duration is the bench line's `roofline.launch_us`
BENCH="$ROOT/bench.py --headline-only --no-setup --no-setup-kv --no-cpu-baseline --no-host-path --no-read-ceiling --no-live-traffic --steps 20 --warmup 5 $*"
# ... and the default run's other sections (fused batches, lone launches, the real database, the host path) in a trace of their own
FULL_BENCH="$ROOT/bench.py --no-setup --no-cpu-baseline --no-read-ceiling --no-live-traffic --other-configs none --steps 10 --warmup 2 $*"
SETUP_BENCH="$ROOT/bench.py --no-cpu-baseline --no-host-path --no-read-ceiling --no-live-traffic --other-configs none --steps 2 --warmup 1 $*"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 $BENCH > "$OUT/trace_bench.json" 2> "$OUT/trace.err"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o fetch -- python3 $BENCH > "$OUT/fetch_bench.json" 2> "$OUT/fetch.err"
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o write -- python3 $BENCH > "$OUT/write_bench.json" 2> "$OUT/write.err"
if [ "${TRAFFIC_ONLY:-0}" != "1" ]; then  # (TRAFFIC_ONLY=1: the headline loop's kernel trace and the two counter passes, nothing else -- the other configs' traffic records)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/full_trace" -o full -- python3 $FULL_BENCH > "$OUT/full_bench.json" 2> "$OUT/full.err"
# the offline kernels (hint matmul, transpose+pack) inside one Server::setup + the setup_roofline timing of each kernel alone
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/setup_trace" -o setup -- python3 $SETUP_BENCH > "$OUT/setup_bench.json" 2> "$OUT/setup.err"
fi
cd "$ROOT" && python3 scripts/summarize_rocprof.py "$OUT" "$TAG" "$HEAD" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
# keep the merge-back small: drop the raw per-dispatch traces except the stats/counter CSVs
find "$OUT" -name "*.db" -delete 2>/dev/null
find "$OUT" -name "*kernel_trace.csv" -size +2M -delete 2>/dev/null
du -sh "$OUT"
