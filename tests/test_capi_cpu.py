"""CPU-side checks of the product library: it loads, exports every symbol include/chalamet_hip.h declares, its host-only
pieces (shape arithmetic, XOF, KV encoder) agree with the oracle, and it fails loudly -- never silently on a CPU path --
when there is no HIP device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from _cases import cf_of


def header_symbols():
    from chalametpir_amd import _native

    text = open(_native.HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cpir_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_header_symbol(native):
    from chalametpir_amd import _native

    syms = header_symbols()
    assert len(syms) >= 40
    raw = C.CDLL(_native.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), f"{s} declared in include/chalamet_hip.h but not exported"
    # and the Python binding types every one of them
    assert set(syms) == set(_native.SIGNATURES), set(syms) ^ set(_native.SIGNATURES)
    assert native.cpir_version().decode().startswith("chalamet_hip")
    assert native.cpir_strerror(5).decode().startswith("The dimensions are incompatible")


def test_shape_helpers_match_oracle(native, orc):
    import chalametpir_amd as cp

    for n in (1, 2, 3, 10, 255, 256, 1000, 1 << 16, 1 << 18, 10 ** 6, 1 << 20, 1 << 22):
        assert cp.find_encoded_db_matrix_element_bit_length(n) == orc.find_encoded_db_matrix_element_bit_length(n)
        for arity in (3, 4):
            assert cp.filter_shape(arity, n) == orc.bff_shape(arity, n)
    for b in range(4, 15):
        assert native.cpir_compression_factor(b) == orc.compression_factor(b) == cf_of(b)
        for v in (1, 64, 1024, 8192):
            assert cp.encoded_num_cols(v, b) == orc.encoded_num_cols(v, b)
    for b in (0, 3, 15, 99):
        assert native.cpir_compression_factor(b) == 0
    with pytest.raises(cp.ChalametPIRError) as e:
        cp.find_encoded_db_matrix_element_bit_length(1 << 46)
    assert e.value.variant == "KVDatabaseSizeTooLarge"
    with pytest.raises(cp.ChalametPIRError) as e:
        cp.filter_shape(5, 10)
    assert e.value.variant == "UnsupportedArityForBinaryFuseFilter"


def test_baseline_config_shapes(native):
    """the five BASELINE configs"""
    import chalametpir_amd as cp

    table = {(1 << 16, 3, 1024): (10, 77_824, 846), (1 << 20, 3, 1024): (9, 1_179_648, 940), (1 << 20, 4, 1024): (9, 1_130_496, 940),
             (1 << 22, 3, 1024): (9, 4_718_592, 940), (1 << 20, 3, 8192): (9, 1_179_648, 7312)}
    for (n, arity, vb), (b, N, Cc) in table.items():
        assert cp.find_encoded_db_matrix_element_bit_length(n) == b
        assert cp.filter_shape(arity, n)[2] == N
        assert cp.encoded_num_cols(vb, b) == Cc
        ref = cp.dtc_layout_for(N, Cc, b, packing=0)
        assert (ref.packing, ref.fields_per_word, ref.chunk_words) == (0, 3, 1024)
        assert ref.words_per_row == -(-N // 3) and ref.words_per_row_padded % ref.chunk_words == 0 and ref.rows_padded % 16 == 0
        assert ref.total_words == ref.rows_padded * ref.words_per_row_padded  # 64-bit: cfg 4/5 exceed 2^32 elements
        # default: planar (exactly b bits per field + one u32 column sum per padded column) for every b >= 9
        L = cp.dtc_layout_for(N, Cc, b)
        ks = -(-N // 512)
        assert (L.packing, L.slots_per_chunk, L.chunk_words, L.words_per_row) == (2, 512, b * 256, -(-N // 3))
        assert L.rows_padded % 16 == 0 and L.words_per_row_padded == ks * b * 16
        assert L.total_words == L.rows_padded * L.words_per_row_padded + L.rows_padded
        assert abs(L.total_words / ref.total_words - b / (32 / 3)) < 0.03  # b bits per field against 32/3 (padding differs)
        cp.tuning_set("layout.planar", 0)
        try:
            D = cp.dtc_layout_for(N, Cc, b)
        finally:
            cp.tuning_set("layout.planar", 1)
        if b == 9:  # dense64: 7 fields per u64 -> 6/7 of the reference packing's bytes
            assert (D.packing, D.fields_per_word, D.slots_per_chunk) == (1, 7, 7168)
            assert 0.85 < D.total_words / ref.total_words < 0.87
        else:  # b = 10: 6 fields per u64 either way, the reference packing stays
            assert D.packing == 0


def test_layout_invariants(native):
    import chalametpir_amd as cp

    for N, Cc, b in ((1, 1, 4), (7, 3, 9), (3 * 1024, 16, 10), (3 * 1024 + 1, 17, 10), (12345, 999, 13)):
        L = cp.dtc_layout_for(N, Cc, b, packing=0)
        cf = cf_of(b)
        assert (L.num_slots, L.num_cols, L.mat_elem_bit_len, L.compression_factor) == (N, Cc, b, cf)
        assert L.words_per_row == -(-N // cf) <= L.words_per_row_padded < L.words_per_row + 1024
        assert Cc <= L.rows_padded < Cc + 16
    for b in range(4, 15):  # planar is offered, and the default, for every bit length: b bits per field for b >= 9, a byte below
        P = cp.dtc_layout_for(50_000, 10, b)
        assert (P.packing, P.chunk_words, P.slots_per_chunk) == (2, max(b, 8) * 256, 512)
        assert cp.dtc_layout_for(50_000, 10, b, packing=2).total_words == P.total_words
    for b in (0, 3, 15):
        with pytest.raises(cp.ChalametPIRError):
            cp.dtc_layout_for(50_000, 10, b, packing=2)
    cp.tuning_set("layout.planar", 0)
    for b in range(4, 15):  # without planar: dense64 is offered exactly for b in {7, 9, 11, 12}
        L = cp.dtc_layout_for(50_000, 10, b)
        if b in (7, 9, 11, 12):
            K = 64 // b
            assert (L.packing, L.fields_per_word, L.chunk_words, L.slots_per_chunk) == (1, K, 2048, K * 1024)
            assert L.words_per_row_padded == -(-50_000 // (K * 1024)) * 2048
        else:
            assert L.packing == 0
            with pytest.raises(cp.ChalametPIRError):
                cp.dtc_layout_for(50_000, 10, b, packing=1)
    cp.tuning_set("layout.dense", 0)
    try:
        assert cp.dtc_layout_for(50_000, 10, 9).packing == 0
    finally:
        cp.tuning_set("layout.dense", 1)
        cp.tuning_set("layout.planar", 1)
    for args, variant in (((0, 3, 9), "InvalidMatrixDimension"), ((5, 0, 9), "InvalidMatrixDimension"),
                          ((5, 3, 3), "ImpossibleEncodedDBMatrixElementBitLength")):
        with pytest.raises(cp.ChalametPIRError) as e:
            cp.dtc_layout_for(*args)
        assert e.value.variant == variant


def test_product_xof_matches_rfc9861_and_oracle(native, orc):
    import chalametpir_amd as cp

    seed = bytes(range(32))
    A = cp.generate_from_seed(5, 1000, seed)
    assert A.tobytes() == orc.turboshake128(seed, 5 * 1000 * 4)
    assert list(A.reshape(-1)[:4]) == [1248867316, 2142359906, 3917524437, 3172935866]
    # RFC 9861 through the product implementation: generate_from_seed(seed) = TurboSHAKE128(seed, D=0x1F)
    z = cp.generate_from_seed(1, 8, bytes(32))
    assert z.tobytes() == orc.turboshake128(bytes(32), 32)
    with pytest.raises(cp.ChalametPIRError) as e:
        cp.generate_from_seed(0, 3, seed)
    assert e.value.variant == "InvalidMatrixDimension"


@pytest.mark.parametrize("arity", [3, 4])
def test_product_encoder_matches_oracle(arity, native, orc):
    """Matrix::from_kv_database on the host (no GPU needed): same D and the same 68 filter bytes as the oracle for the same
    key order and candidate seeds, including seeds that fail and force further attempts"""
    import chalametpir_amd as cp

    rng = np.random.default_rng(600 + arity)
    for n in (1, 2, 5, 64, 1000, 20000):
        keys = list({rng.bytes(int(rng.integers(1, 33))): 0 for _ in range(n)})
        vals = [rng.bytes(int(rng.integers(1, 100))) for _ in keys]
        b = int(rng.integers(4, 15))
        fseeds = rng.bytes(3200)
        D1, f1 = cp.encode_kv_database(dict(zip(keys, vals)), arity, b, fseeds)
        D2, f2, used = orc.from_kv_database(arity, keys, vals, b, fseeds)
        assert np.array_equal(D1, D2) and f1 == f2.to_bytes(), (n, b, used)
        for k, v in list(zip(keys, vals))[:20]:
            assert orc.recover_value(D1, orc.Filter.from_bytes(f1), k) == v
    with pytest.raises(cp.ChalametPIRError) as e:
        cp.encode_kv_database({}, arity, 8)
    assert e.value.variant == "EmptyKVDatabase"
    # OS-seeded construction (filter_seed_material=None) still yields a decodable matrix
    keys = [bytes([i]) * 16 for i in range(50)]
    D, fb = cp.encode_kv_database({k: k[:3] for k in keys}, arity, 9)
    filt = orc.Filter.from_bytes(fb)
    assert all(orc.recover_value(D, filt, k) == k[:3] for k in keys)


def test_no_device_fails_loudly(native):
    """Without a GPU nothing computes: Device() raises HipNoDevice; there is no CPU fallback to fall into."""
    import torch

    import chalametpir_amd as cp

    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    assert cp.Device.count() == 0
    with pytest.raises(cp.ChalametPIRError) as e:
        cp.Device(0)
    assert e.value.variant == "HipNoDevice"
    with pytest.raises(cp.ChalametPIRError):
        cp.Server.setup(bytes(32), {b"k": b"v"}, 3)
    with pytest.raises(cp.ChalametPIRError):
        cp.Server.setup_from_matrix(bytes(32), np.ones((4, 4), dtype=np.uint32), 9)


def test_null_handles_are_refused_not_dereferenced(native):
    """entry points that only look at a handle say CPIR_ERR_INVALID_ARGUMENT for a NULL one (no GPU needed to find that out)"""
    import ctypes as C

    out64 = (C.c_uint64 * 8)()
    outd = (C.c_double * 8)()
    assert native.cpir_server_host_path_counts(None, out64) == 68  # CPIR_ERR_INVALID_ARGUMENT
    assert native.cpir_server_setup_timings(None, outd) == 68
    q = (C.c_uint32 * 4)()
    assert native.cpir_server_respond(None, q, 1, 4, q) == 68


def test_product_never_imports_the_oracle():
    """the oracle is test infrastructure: nothing under chalametpir_amd/ may import, link or call it"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "chalametpir_amd")
    for dirpath, _, files in os.walk(pkg):
        if os.sep + "lib" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "chalamet_oracle" not in text and "from oracle" not in text and "import oracle" not in text, os.path.join(dirpath, f)


def test_every_documented_tuning_key_is_accepted_and_bounded():
    """the keys include/chalamet_hip.h documents for cpir_tuning_set exist (host-side state: no GPU needed), reject values outside their
    range, and every key the library accepts is documented in the header"""
    import re

    import chalametpir_amd as cp
    from chalametpir_amd.errors import ChalametPIRError

    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(ROOT, "include", "chalamet_hip.h")).read()
    doc = header[header.index("Tuning knobs of the respond kernel"):header.index("int cpir_tuning_set")]
    documented = set(re.findall(r'"((?:respond|matmul|layout|pack)\.[a-z_]+)"', doc))
    source = open(os.path.join(ROOT, "chalametpir_amd", "csrc", "respond.hip")).read()
    source = re.sub(r"#ifdef CPIR_DIAG.*?#endif", "", source, flags=re.S)  # (the diagnosis build's own key is not in the release library)
    accepted = set(re.findall(r'!strcmp\(key, "([a-z_.]+)"\)', source))
    assert accepted == documented, (sorted(accepted - documented), sorted(documented - accepted))
    defaults = {"respond.ks_major": 1, "respond.host_zero_copy": 1, "respond.host_fill_timeout_us": 2000, "respond.batch_fusion": 1,
                "respond.interleave_passes": -1, "matmul.mfma": 1}
    for key, value in defaults.items():
        cp.tuning_set(key, value)
    for key, bad in (("respond.ks_major", 4), ("respond.ks_major", 0), ("respond.host_fill_timeout_us", -5),
                     ("respond.host_fill_timeout_us", 2_000_000), ("respond.inplace_seats", 1), ("respond.inplace_seats", 5),
                     ("respond.no_such_key", 1)):
        with pytest.raises(ChalametPIRError):
            cp.tuning_set(key, bad)


def test_the_suite_registers_only_mappings_of_its_own():
    """regression guard for the abort of the GPU suite (profiles/HISTORY_design_r3.md 4.6): hipHostRegister over glibc HEAP memory (a numpy array) is followed,
    on the GPU boxes, by a GPU memory fault some allocations later -- with the runtime alone.  Every hipHostRegister of this suite must
    therefore go through tests/_cases.py::OwnMapping (a private mapping, registered, unregistered with the return code checked, unmapped)."""
    import re

    here = os.path.dirname(os.path.abspath(__file__))
    for name in sorted(os.listdir(here)):
        if not name.endswith(".py") or name == os.path.basename(__file__):
            continue
        text = open(os.path.join(here, name)).read()
        for m in re.finditer(r"cudaHost(?:Un)?[Rr]egister\(([^,)]*)", text):
            assert m.group(1).strip() == "own.address", (name, m.group(0))
        assert "hipHostRegister(" not in text or name == "_cases.py" or "OwnMapping" in text, name


def test_host_gather_variants_agree(native):
    """the host routine that compacts a lone query (host_gather.cpp): whatever variant this CPU runs equals numpy's take, on ragged
    counts around the vector widths"""
    import chalametpir_amd as cp

    rng = np.random.default_rng(5)
    src = rng.integers(0, 1 << 32, size=100003, dtype=np.uint64).astype(np.uint32)
    for count in (0, 1, 7, 8, 15, 16, 17, 31, 33, 1000, 65536, 90001):
        idx = np.sort(rng.choice(src.size, size=count, replace=False)).astype(np.uint32)
        assert np.array_equal(cp.host_gather(src, idx), src[idx]), count
    assert cp.host_gather_variant() in ("scalar", "avx2", "avx512", "avx512-compress")
    # the streaming form: a bitmap over the source, any [s_lo, s_hi) window (copy jobs start and end at arbitrary source words)
    for n in (1, 15, 16, 17, 1000, 4096, 4607, 9000, 100003):  # (around the streaming form's 4 096-word buffer too)
        keep = rng.random(n) < 0.89
        assert np.array_equal(cp.host_compress(src[:n], keep), src[:n][keep]), n
        assert np.array_equal(cp.host_compress(src[:n], keep, dst_misalign_words=3), src[:n][keep]), n  # (the plain form: an unaligned destination)
        for _ in range(4):
            lo = int(rng.integers(0, n))
            hi = int(rng.integers(lo, n + 1))
            assert np.array_equal(cp.host_compress(src[:n], keep, lo, hi), src[lo:hi][keep[lo:hi]]), (n, lo, hi)
    assert cp.host_compress(src[:64], np.zeros(64, dtype=bool)).size == 0 and np.array_equal(cp.host_compress(src[:64], np.ones(64, dtype=bool)), src[:64])


def test_how_a_fused_batch_is_cut_into_passes():
    """cpir_respond_batch_pass_width (host-side arithmetic, no GPU): on the planar packing as few passes as 24 queries each allow, all of
    about the same width; passes of 4 where respond.ks_major sends fused passes to the step-major kernel; never more than the batch; other
    packings 4; every cut covers the batch with passes of at most that width plus a remainder below it"""
    import chalametpir_amd as cp

    try:
        planar = cp.dtc_layout_for(1179648, 940, 9)
        assert int(planar.packing) == 2
        widths = {k: cp.respond_batch_pass_width(planar, k) for k in (1, 4, 5, 12, 13, 24, 25, 32, 47, 48, 49, 72, 96, 100, 241, 1009)}
        assert widths == {1: 1, 4: 4, 5: 5, 12: 12, 13: 13, 24: 24, 25: 13, 32: 16, 47: 24, 48: 24, 49: 17, 72: 24, 96: 24, 100: 20, 241: 22, 1009: 24}
        for k, w in widths.items():
            assert 1 <= w <= min(k, 24) and (k // w) * w + (k % w) == k and -(-k // w) <= -(-k // 24) + 1
        small = cp.dtc_layout_for(147456, 940, 9)  # a 1/8 shard: the same cut (the wide kernel takes interleaved launches too)
        assert cp.respond_batch_pass_width(small, 48) == 24 and cp.respond_batch_pass_width(small, 12) == 12
        cp.tuning_set("respond.ks_major", 2)
        assert [cp.respond_batch_pass_width(planar, k) for k in (1, 3, 4, 5, 12, 48)] == [1, 3, 4, 4, 4, 4]
        cp.tuning_set("respond.ks_major", 1)
        cp.tuning_set("layout.planar", 0)
        other = cp.dtc_layout_for(1179648, 940, 9)
        assert int(other.packing) != 2 and cp.respond_batch_pass_width(other, 48) == 4 and cp.respond_batch_pass_width(other, 3) == 2
        assert cp.respond_batch_pass_width(planar, 0) == 0
    finally:
        cp.tuning_reset()


def test_release_library_has_no_wrong_answer_switches(native):
    """Server::respond has no mode in which it lies (server.rs:184-190): the ablation switches and the timing traces of the tuning scripts
    are compiled only into the diagnosis build (-DCPIR_DIAG, `make diag`).  The release library neither contains the names of the
    environment variables nor accepts the tuning key, and its only getenv()s are the documented, result-neutral ones."""
    import re
    import subprocess

    import chalametpir_amd as cp
    from chalametpir_amd import _native
    from chalametpir_amd.errors import ChalametPIRError

    text = subprocess.run(["strings", "-a", _native.LIB_PATH], capture_output=True, text=True, check=True).stdout
    for name in ("CPIR_WIDE_ABLATE", "CPIR_KS_TRACE", "matmul.ablate", "[ks trace]", "[wide trace]"):
        assert name not in text, name
    for key in ("matmul.ablate", "respond.ablate", "respond.wide_ablate"):
        with pytest.raises(ChalametPIRError):
            cp.tuning_set(key, 1)
    # every environment variable the release sources read is on this list, and none of them changes a result
    allowed = {"CPIR_ABORT_BACKTRACE", "CPIR_RESPOND_TRACE", "CPIR_GATHER", "CPIR_XOF_SCALAR"}
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "chalametpir_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if not name.endswith((".hip", ".cpp", ".hpp")):
            continue
        src = open(os.path.join(csrc, name)).read()
        src = re.sub(r"#ifdef CPIR_DIAG.*?#endif", "", src, flags=re.S)  # (the diagnosis build's own)
        for var in re.findall(r'getenv\("([A-Z_0-9]+)"\)', src):
            assert var in allowed, (name, var)


def test_product_encoder_fuzz_against_the_oracle(native, orc):
    """Matrix::from_kv_database on shapes the reference's own tests do not draw (utils.rs:22-45 draws keys of 16-32 and values of 1-512 bytes):
    keys of 0..79 bytes, values of 0..299 bytes, 1..3000 pairs, every bit length, both arities -- same D, same 68 filter bytes, the same
    refusals as the oracle; and the reference's own quirk holds: an EMPTY value encodes, but its row is RowNotDecodable (the boundary mark
    must lie beyond byte 32: serialization.rs:169), in the product's matrix exactly as in the oracle's"""
    import chalametpir_amd as cp

    rng = np.random.default_rng(4242)
    empty_values_seen = 0
    for case in range(80):
        arity = int(rng.integers(3, 5))
        n = int([rng.integers(1, 6), rng.integers(1, 200), rng.integers(200, 3000)][int(rng.integers(0, 3))])
        keys = list({rng.bytes(int(rng.integers(0, 80))): 0 for _ in range(n)})
        vals = [rng.bytes(int(rng.integers(0, 300))) for _ in keys]
        b = int(rng.integers(4, 15))
        fseeds = rng.bytes(3200)
        D1, f1 = cp.encode_kv_database(dict(zip(keys, vals)), arity, b, fseeds)
        D2, f2, _ = orc.from_kv_database(arity, keys, vals, b, fseeds)
        assert np.array_equal(D1, D2) and f1 == f2.to_bytes(), (case, n, b, arity)
        filt = orc.Filter.from_bytes(f1)
        for k, v in list(zip(keys, vals))[:6]:
            if len(v) == 0:
                empty_values_seen += 1
                with pytest.raises(Exception) as e:
                    orc.recover_value(D1, filt, k)
                assert "11" in str(e.value)  # RowNotDecodable (error.rs:37)
            else:
                assert orc.recover_value(D1, filt, k) == v, (case, len(k), len(v))
    assert empty_values_seen > 0
