"""ASan + UBSan over the product's host-side C++ (XOF, shapes, KV encoder) and over the oracle's C.  GPU AddressSanitizer is
not available on the pool, so the sanitised surface is the CPU code; the kernels are covered by the host-side shape checks in
front of every launch plus the parity tests."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "chalametpir_amd", "csrc")


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_product_host_code_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "host_sanitize")
    srcs = [os.path.join(ROOT, "tests", "native", "host_sanitize.cpp")] + [os.path.join(CSRC, f) for f in
                                                                            ("host_xof.cpp", "host_shapes.cpp", "host_encoder.cpp")]
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
           "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__=1", "-pthread", "-o", exe] + srcs
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-4000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0 and "host sanitizer run ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_oracle_under_asan_ubsan(tmp_path):
    """the oracle itself (what every parity claim rests on) through the same property checks under sanitizers"""
    exe = str(tmp_path / "oracle_sanitize")
    drv = tmp_path / "drv.c"
    drv.write_text(r'''
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "chalamet_oracle.h"
int main(void) {
  for (unsigned b = 4; b <= 14; b++) {
    unsigned cf = or_compression_factor(b);
    for (unsigned tail = 0; tail < cf; tail++) {
      uint64_t N = cf * 37 + tail, C = 9, W = (N + cf - 1) / cf;
      uint32_t *D = malloc(N * C * 4), *Dt = malloc(N * C * 4), *dtc = malloc(C * W * 4), *q = malloc(N * 4), *r = malloc(C * 4), *r2 = malloc(C * 4);
      or_synth_fill_u32(D, N * C, b, 0, (1u << b) - 1); or_synth_fill_u32(q, N, 99, 0, 0xffffffffu);
      if (or_transpose(D, N, C, Dt) || or_row_wise_compress(Dt, C, N, b, dtc)) return 1;
      if (or_row_vector_x_compressed_transposed_matrix(q, 1, N, dtc, C, W, N, b, r)) return 2;
      if (or_mul(q, 1, N, D, N, C, r2) || memcmp(r, r2, C * 4)) return 3;
      uint32_t* back = malloc(N * C * 4);
      if (or_row_wise_decompress(dtc, C, W, b, N, back) || memcmp(back, Dt, N * C * 4)) return 4;
      free(D); free(Dt); free(dtc); free(q); free(r); free(r2); free(back);
    }
  }
  uint8_t key[5] = {1,2,3,4,5}, val[40]; memset(val, 0xA5, sizeof val);
  for (unsigned b = 4; b <= 14; b++) {
    uint64_t cols = or_encoded_num_cols(sizeof val, b);
    uint32_t* row = malloc(cols * 4); uint8_t out[256]; size_t n = 0;
    or_encode_kv_as_row(key, 5, val, sizeof val, b, cols, row);
    if (or_decode_kv_from_row(row, cols, b, out, sizeof out, &n) || n != 32 + sizeof val || memcmp(out + 32, val, sizeof val)) return 5;
    free(row);
  }
  puts("oracle sanitizer run ok");
  return 0;
}
''')
    cmd = ["gcc", "-std=gnu11", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fopenmp",
           "-I" + os.path.join(ROOT, "oracle"), "-o", exe, str(drv), os.path.join(ROOT, "oracle", "chalamet_oracle.c"), "-lm"]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-4000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(os.environ, OMP_NUM_THREADS="4", ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode == 0 and "oracle sanitizer run ok" in r.stdout, str(r.returncode) + r.stdout[-2000:] + r.stderr[-4000:]
