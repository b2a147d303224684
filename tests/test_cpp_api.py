"""include/chalamet_hip.hpp -- the reference's `Server` API surface in C++ (the compiled counterpart of the source-only Rust shim,
rust/server_hip.rs) -- compiled here, and driven on the GPU through tests/native/test_pir.cpp, which reads like the reference's own
integration test (integrations/src/test_pir.rs:12-142): random key-value databases, Server::setup::<ARITY>, ten keys queried per
database through the client restatement of the oracle (the checker), every value recovered."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "chalametpir_amd", "lib")
ORACLE = os.path.join(ROOT, "oracle")


def compile_cpp(src, exe, extra=()):
    from chalametpir_amd import _native
    from oracle import oracle as orc  # checker only (the client side of the end-to-end test)

    _native.load()
    orc.lib()  # (builds oracle/_build/libchalamet_oracle.so if it is not there)
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), "-I" + ORACLE, src, "-o", exe,
           "-L" + LIB, "-lchalamet_hip", "-L" + os.path.join(ORACLE, "_build"), "-lchalamet_oracle", "-Wl,-rpath," + LIB,
           "-Wl,-rpath," + os.path.join(ORACLE, "_build"), "-Wl,-rpath-link,/opt/rocm/lib", "-fopenmp", "-pthread", *extra]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-4000:]


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_cpp_mirror_compiles_and_maps_errors_without_a_device(tmp_path):
    """the header under -Wall -Wextra -Werror; without a GPU Server::setup is HipDeviceNotFound (the product has no CPU fallback), an empty
    database is EmptyKVDatabase before any device is asked for (server.rs:48-51), an unsupported arity does not compile (matrix.rs:638)"""
    import torch

    src = tmp_path / "api.cpp"
    src.write_text(r'''
#include <unordered_map>
#include "chalamet_hip.hpp"
using namespace chalametpir;
int main() {
  const std::unordered_map<Bytes, Bytes, BytesHash> empty;
  auto e = Server::setup<3>(std::array<uint8_t, 32>{}, empty);
  if (!e.is_err() || e.unwrap_err().kind != ChalametPIRError::EmptyKVDatabase) return 1;
  const std::vector<uint8_t> k{1, 2, 3}, v{4, 5};
  std::unordered_map<Bytes, Bytes, BytesHash> one;
  one.emplace(Bytes(k), Bytes(v));
  auto r = Server::setup<4>(std::array<uint8_t, 32>{}, one);
  if (r.is_ok()) return 0;                                   // (a device is present: fine)
  return r.unwrap_err().kind == ChalametPIRError::HipDeviceNotFound && r.unwrap_err().to_string().size() > 0 ? 0 : 2;
}
''')
    exe = str(tmp_path / "api")
    compile_cpp(str(src), exe)
    p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, (p.returncode, p.stderr[-2000:])
    if not torch.cuda.is_available():
        bad = tmp_path / "bad.cpp"
        bad.write_text('#include <unordered_map>\n#include "chalamet_hip.hpp"\nint main() { std::unordered_map<chalametpir::Bytes, chalametpir::Bytes, chalametpir::BytesHash> m; '
                       'return chalametpir::Server::setup<5>(std::array<uint8_t, 32>{}, m).is_ok(); }\n')
        b = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"), str(bad)], capture_output=True, text=True, timeout=120)
        assert b.returncode != 0 and "ARITY == 3 || ARITY == 4" in b.stderr


@pytest.mark.gpu
def test_keyword_pir_end_to_end_through_the_cpp_mirror(tmp_path):
    """integrations/src/test_pir.rs in C++: both arities, databases of 2^8 .. 2^16 pairs with keys of 16-32 and values of 1-512 bytes, ten keys
    each, a cloned Server answering every other query, malformed and mis-sized queries rejected with the reference's errors"""
    exe = str(tmp_path / "test_pir")
    compile_cpp(os.path.join(ROOT, "tests", "native", "test_pir.cpp"), exe)
    p = subprocess.run([exe, "5", "16"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "test_pir ok" in p.stdout, p.stdout[-3000:] + p.stderr[-3000:]
    assert p.stdout.count("test_keyword_pir_with_3_wise_xor_filter") == 5 and p.stdout.count("test_keyword_pir_with_4_wise_xor_filter") == 5
