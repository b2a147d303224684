#!/bin/bash
# Runs ON THE GPU BOX: the round's long-running guards at the library as it stands -- ThreadSanitizer driver (make -C chalametpir_amd/csrc tsan
# beforehand; N runs), the two soaks (full database / 11 % empty rows) and the lifecycle soak.  usage: scripts/probes/round_soaks.sh <tag> [tsan runs [soak seconds]]
set -u
TAG=${1:-r6}; RUNS=${2:-5}; SECS=${3:-100}
O=gpurun_out/$TAG; mkdir -p $O
: > $O/tsan_runs.txt
for i in $(seq 1 $RUNS); do
  TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0" timeout -k 10 300 chalametpir_amd/lib/tsan/tsan_driver > $O/tsan_$i.out 2> $O/tsan_$i.err; rc=$?
  echo "run $i: exit $rc, $(grep -c 'WARNING: ThreadSanitizer' $O/tsan_$i.err) reports, $(grep -h 'mismatches' $O/tsan_$i.out | tail -1)" | tee -a $O/tsan_runs.txt
done
python3 scripts/summarize_tsan.py $O/tsan_$RUNS.err > $O/tsan_summary.txt 2>&1; tail -8 $O/tsan_summary.txt
tail -12 $O/tsan_$RUNS.out >> $O/tsan_summary.txt
timeout -k 10 $((SECS + 200)) python3 scripts/soak.py --seconds $SECS > $O/soak_plain.txt 2>&1; echo "soak plain rc $?"; tail -3 $O/soak_plain.txt
timeout -k 10 $((SECS + 200)) python3 scripts/soak.py --seconds $SECS --holes 0.11 > $O/soak_compacted.txt 2>&1; echo "soak compacted rc $?"; tail -3 $O/soak_compacted.txt
timeout -k 10 300 python3 scripts/lifecycle_soak.py --seconds 60 > $O/lifecycle_soak.txt 2>&1; echo "lifecycle rc $?"; tail -3 $O/lifecycle_soak.txt
rm -f $O/tsan_[0-9]*.err
