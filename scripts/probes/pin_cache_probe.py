#!/usr/bin/env python3
"""Does the HIP runtime report PAGEABLE host memory as page-locked?  (GPU box.)

The lone-caller path of cpir_server_respond used to ask hipPointerGetAttributes whether a caller's query buffer is page-locked and, on a
yes, let the kernel read it in place.  This probe shows where a yes can come from without the caller ever having registered anything: a
pageable H2D copy of the runtime itself (here torch's `.cuda()` of a numpy array) pins the source range internally and keeps the pinning in
a cache; a later, unrelated pageable buffer that glibc places at the same heap addresses is then reported as hipMemoryTypeHost with a
device pointer -- until the runtime drops the cached pinning, at a moment of its own choosing."""
import ctypes as C
import gc
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from chalametpir_amd import _native  # noqa: E402

_native.load()  # (loads torch's bundled HIP runtime first, as the product does)
hip = C.CDLL("libamdhip64.so", mode=C.RTLD_GLOBAL) if False else C.CDLL(None)


class Attr(C.Structure):  # hipPointerAttribute_t (ROCm 6/7)
    _fields_ = [("type", C.c_int), ("device", C.c_int), ("devicePointer", C.c_void_p), ("hostPointer", C.c_void_p), ("isManaged", C.c_int),
                ("allocationFlags", C.c_uint)]


def attributes(addr):
    a = Attr()
    fn = hip.hipPointerGetAttributes
    fn.argtypes = [C.POINTER(Attr), C.c_void_p]
    fn.restype = C.c_int
    e = fn(C.byref(a), C.c_void_p(addr))
    return e, a.type, a.devicePointer


torch.zeros(1, device="cuda")
names = {0: "unregistered", 1: "HOST (page-locked)", 2: "device", 3: "array/unified", 4: "managed"}
for mb in (1, 4, 24, 64, 200):
    n = mb * (1 << 20) // 4
    src = np.ones(n, dtype=np.int32)  # pageable; below glibc's (grown) mmap threshold this lives in the brk heap
    addr = src.ctypes.data
    before = attributes(addr + 4096)
    t = torch.from_numpy(src).cuda()
    torch.cuda.synchronize()
    after = attributes(addr + 4096)
    del t, src
    gc.collect()
    again = np.zeros(n // 2, dtype=np.int32)  # an unrelated, never registered buffer -- often at the same addresses
    reused = again.ctypes.data
    later = attributes(reused + 4096)
    print(f"{mb:4d} MB pageable array at {addr:#x}: before the copy {names.get(before[1], before[1])} (rc {before[0]}); after torch's .cuda() "
          f"{names.get(after[1], after[1])} devptr {after[2] or 0:#x}; a NEW array at {reused:#x} ({'same' if reused == addr else 'other'} address): "
          f"{names.get(later[1], later[1])} devptr {later[2] or 0:#x}", flush=True)
    del again
