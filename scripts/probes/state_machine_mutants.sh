#!/bin/bash
# Does the GPU-free test of the host state machine (tests/test_host_state_machine.py) have teeth?  Three seeded bugs in a COPY of
# host_respond.hip, each built against the simulated runtime under ThreadSanitizer and run for 15 000 calls: every one must fail.
#   m1  a follower of an uploaded round leaves as soon as its arena is LAUNCHED (reads the responses before they are there)
#   m2  a lone polled launch with the staging helpers announces one copy job more than has been copied
#   m3  an arena is freed by the first caller out instead of the last
# usage: scripts/probes/state_machine_mutants.sh   (CPU only, ~2 minutes; prints one line per mutant)
set -u
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
W=$(mktemp -d /tmp/cpir_mut_XXXX)
cp $ROOT/chalametpir_amd/csrc/*.hip $ROOT/chalametpir_amd/csrc/*.hpp $ROOT/chalametpir_amd/csrc/*.cpp $W/
sed -i "s#\"../../include/chalamet_hip.h\"#\"$ROOT/include/chalamet_hip.h\"#" $W/cpir_internal.hpp
python3 - $W <<'PY'
import sys
w = sys.argv[1]
s = open(w + "/host_respond.hip").read()
def mutant(name, old, new):
    assert old in s, name
    open(f"{w}/{name}.hip", "w").write(s.replace(old, new))
mutant("m1", "    srv->cv.wait(lk, [&] { return a->state == RespondArena::DONE; });\n    if (tr) srv->trace.ns_follow",
       "    srv->cv.wait(lk, [&] { return a->state == RespondArena::DONE || a->state == RespondArena::LAUNCHED; });\n    if (tr) srv->trace.ns_follow")
mutant("m2", "          publish_fill_progress(a->fill_progress, i + 1 == n_jobs ? 0xffffffffu : (uint32_t)((i + 1) * kStepsPerJob));\n        }\n      } else if (rc == CPIR_OK) {",
       "          publish_fill_progress(a->fill_progress, i + 1 == n_jobs ? 0xffffffffu : (uint32_t)((i + 2) * kStepsPerJob));\n        }\n      } else if (rc == CPIR_OK) {")
mutant("m3", "  if (++a->left == a->joined) {  // last one out frees the arena\n    a->state = RespondArena::FREE;\n    a->joined = a->staged = a->left = 0;\n    srv->cv.notify_all();\n  }\n  return status;\n}\n\nint cpir_server_respond_bytes",
       "  if (++a->left >= 1) {  // MUTANT\n    a->state = RespondArena::FREE;\n    a->joined = a->staged = a->left = 0;\n    srv->cv.notify_all();\n  }\n  return status;\n}\n\nint cpir_server_respond_bytes")
PY
CL=/opt/rocm/lib/llvm/bin/clang++
for m in m1 m2 m3; do
  $CL -std=c++17 -O1 -g -fsanitize=thread -pthread -I$ROOT/tests/native/sim_hip -I$W -x c++ $W/$m.hip $W/host_gather.cpp $ROOT/tests/native/sim_hip/sim_runtime.cpp $ROOT/tests/native/host_state_machine_driver.cpp -o $W/drv_$m || exit 1
  timeout 300 $W/drv_$m 15000 3 > $W/$m.out 2>&1; rc=$?
  echo "$m: exit code $rc, ThreadSanitizer reports $(grep -c 'WARNING: ThreadSanitizer' $W/$m.out), $(grep -o 'wrong [0-9]*' $W/$m.out | tail -1), last line: $(tail -1 $W/$m.out | cut -c1-160)"
done
rm -rf $W
