#!/usr/bin/env python3
"""Reads the stderr of chalametpir_amd/lib/tsan/tsan_driver (ThreadSanitizer reports) and says which reports concern the library's own host
code: for every access / mutex acquisition of a report, the module of the first frame that is not one of the sanitizer's interceptors; a
report counts against the library if that module is libchalamet_hip.so or the driver for ANY of its stacks.  (The HIP / HSA runtime is not
instrumented: its internal synchronisation is invisible to the sanitizer, and reports between two of its own accesses are its business.)
   usage: summarize_tsan.py <stderr file>"""
import collections
import re
import sys

text = open(sys.argv[1], errors="replace").read()
reports = text.split("WARNING: ThreadSanitizer: ")[1:]
kinds = collections.Counter()
pairs = collections.Counter()
ours = []
for rep in reports:
    kind = rep.split(" (pid")[0].split("\n")[0].strip()
    kinds[kind] += 1
    mods = []
    for block in re.split(r"\n\s*\n", rep):
        head = block.strip().split("\n")[0] if block.strip() else ""
        # only the stacks of the accesses / acquisitions themselves -- not "thread created at", "mutex created at", "location is heap block"
        if not re.match(r"(Read|Write|Previous (read|write|atomic)|Atomic|Mutex M\d+ acquired|Cycle in lock order)", head.strip(), re.I):
            continue
        for m in re.finditer(r"#\d+ (\S+) .*?\(([^+)]+)\+0x[0-9a-f]+\)", block):
            fn, mod = m.group(1), m.group(2)
            if mod.startswith(("libtsan", "libclang_rt")) or fn.startswith("__interceptor") or (fn in ("memcpy", "memset", "memmove", "free", "malloc", "operator", "pthread_mutex_lock", "pthread_mutex_unlock") and "tsan" in mod):
                continue
            if "tsan_driver" in mod and fn in ("memcpy", "memset", "memmove", "malloc", "free", "calloc", "posix_memalign", "operator"):
                continue  # the sanitizer's interceptors are linked into the driver binary
            mods.append(mod.split("/")[-1])
            break
    pairs[tuple(sorted(set(mods)))] += 1
    if any("chalamet" in m or "tsan_driver" in m for m in mods):
        ours.append(rep[:1500])
print(f"{len(reports)} reports: " + ", ".join(f"{v} {k}" for k, v in kinds.most_common()))
print("modules of the first non-interceptor frame of the reported accesses / acquisitions:")
for k, v in pairs.most_common():
    print(f"  {v:5d}  {k}")
print(f"reports that name libchalamet_hip.so or the driver in such a frame: {len(ours)}")
for r in ours[:5]:
    print("-----\n" + r)
