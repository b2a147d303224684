#!/usr/bin/env python3
"""Writes the per-kernel resource table of the CURRENT release build (profiles/r5_kernel_resources.txt) from the compiler's own reports
(chalametpir_amd/lib/obj/*.usage, the files tests/test_kernel_resources.py reads).   python scripts/kernel_resources_report.py > profiles/rN_kernel_resources.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_kernel_resources import _reports  # noqa: E402

rep = _reports()
spilling = sorted(n for n, k in rep.items() if k["scratch"] or k["vgpr_spill"])
print("# Per-kernel resource report of the release build (hipcc -Rpass-analysis=kernel-resource-usage, written next to every object by the Makefile;")
print("# tests/test_kernel_resources.py reads the same files; scripts/kernel_resources_report.py prints this table).")
print(f"# {len(rep)} kernels; columns: VGPRs, AGPRs, scratch bytes per lane, occupancy (waves per SIMD), static LDS bytes per block, source file")
print(f"# kernels with scratch or spilled VGPRs: {len(spilling)} {spilling}")
for name, k in sorted(rep.items(), key=lambda kv: (kv[1]["file"], kv[0])):
    print(f"{k['vgpr']:4d} {k['agpr']:4d} {k['scratch']:4d} {k['occupancy']:2d} {k['lds']:7d}  {k['file']:18s} {name}")
