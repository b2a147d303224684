"""The oracle pinned against every property test the reference holds for this path (the reference has NO fixed vectors;
all of its tests are OS-seeded property tests -- SURVEY.md section 4 / 8c), against RFC 9861 known answers for the
third-party XOF, and against the byte sizes published in the reference README.  Seeded here, so reproducible."""
import numpy as np
import pytest

from _cases import ALL_BITS, cf_of, random_db_matrix, random_query, unwire, wire


# ---- TurboSHAKE128: RFC 9861 test vectors (the reference's `turboshake =0.4.1` dependency is not vendored) --------------
def ptn(n):
    return bytes(i % 0xFB for i in range(n))


RFC9861 = [
    (b"", 0x1F, 32, "1e415f1c5983aff2169217277d17bb538cd945a397ddec541f1ce41af2c1b74c"),
    (b"", 0x1F, 64, "1e415f1c5983aff2169217277d17bb538cd945a397ddec541f1ce41af2c1b74c"
                    "3e8ccae2a4dae56c84a04c2385c03c15e8193bdf58737363321691c05462c8df"),
    (ptn(1), 0x1F, 32, "55cedd6f60af7bb29a4042ae832ef3f58db7299f893ebb9247247d856958daa9"),
    (ptn(17), 0x1F, 32, "9c97d036a3bac819db70ede0ca554ec6e4c2a1a4ffbfd9ec269ca6a111161233"),
    (ptn(17 ** 2), 0x1F, 32, "96c77c279e0126f7fc07c9b07f5cdae1e0be60bdbe10620040e75d7223a624d2"),
    (ptn(17 ** 3), 0x1F, 32, "d4976eb56bcf118520582b709f73e1d6853e001fdaf80e1b13e0d0599d5fb372"),
    (b"\xff\xff\xff", 0x01, 32, "bf323f940494e88ee1c540fe660be8a0c93f43d15ec006998462fa994eed5dab"),
]


@pytest.mark.parametrize("msg,sep,n,want", RFC9861)
def test_turboshake128_rfc9861(msg, sep, n, want, orc):
    assert orc.turboshake128(msg, n, sep).hex() == want


def test_turboshake128_long_output_rfc9861(orc):
    assert orc.turboshake128(b"", 10032)[-32:].hex() == "a3b9b0385900ce761f22aed548e754da10a5242d62e8c658e3f3a923a7555607"


def test_generate_from_seed_is_the_xof_stream(orc):
    """matrix.rs:541-558: A = squeeze(absorb(seed), 0x1F) reinterpreted as row-major LE u32"""
    seed = bytes(range(32))
    A = orc.generate_from_seed(7, 13, seed)
    assert A.tobytes() == orc.turboshake128(seed, 7 * 13 * 4)
    assert list(A.reshape(-1)[:4]) == [1248867316, 2142359906, 3917524437, 3172935866]  # SURVEY.md 8c cross-check
    with pytest.raises(orc.OracleError) as e:
        orc.generate_from_seed(0, 4, seed)
    assert e.value.code == orc.ERR_INVALID_MATRIX_DIMENSION


# ---- sizes published in the reference README (README.md:33-36) pin N, C, b and the wire header ------------------------
def test_readme_byte_sizes(orc):
    n = 1 << 20
    b = orc.find_encoded_db_matrix_element_bit_length(n)
    C = orc.encoded_num_cols(1024, b)
    _, _, N3 = orc.bff_shape(3, n)
    _, _, N4 = orc.bff_shape(4, n)
    assert 8 + 4 * 1774 * C == 6_670_248  # hint
    assert 8 + 4 * N3 == 4_718_600  # query, 3-wise
    assert 8 + 4 * N4 == 4_521_992  # query, 4-wise
    assert 8 + 4 * C == 3_768  # response
    assert len(orc.Filter(bytes(32), 3, 1, 1, 1, 1, 9).to_bytes()) == 68
    assert (b, C, N3, N4) == (9, 940, 1_179_648, 1_130_496)


def test_bit_length_selection(orc):
    """server.rs:193-218: 2^32 >= 8 * 4^b * floor(sqrt(n))"""
    assert [orc.find_encoded_db_matrix_element_bit_length(1 << k) for k in (0, 8, 16, 18, 19, 20, 22, 42)] == [14, 12, 10, 10, 9, 9, 9, 4]
    with pytest.raises(orc.OracleError) as e:
        orc.find_encoded_db_matrix_element_bit_length(1 << 46)
    assert e.value.code == orc.ERR_KV_DATABASE_SIZE_TOO_LARGE
    for n in (1, 2, 3, 999, 65536, 10 ** 6):
        b = orc.find_encoded_db_matrix_element_bit_length(n)
        root = int(np.floor(np.sqrt(n)))
        assert (1 << 32) >= 8 * 4 ** b * root and (1 << 32) < 8 * 4 ** (b + 1) * root


# ---- Matrix property tests --------------------------------------------------------------------------------------------
def test_matrix_multiplication_is_correct(orc):
    """matrix.rs:1275-1317: A*I = A = I*A"""
    rng = np.random.default_rng(1)
    seed = rng.bytes(32)
    for _ in range(20):
        r, c = int(rng.integers(1, 1025)), int(rng.integers(1, 1025))
        A = orc.generate_from_seed(r, c, seed)
        assert np.array_equal(orc.mul(A, orc.identity(c)), A)
        assert np.array_equal(orc.mul(orc.identity(r), A), A)


def test_matrix_multiplication_against_numpy(orc):
    rng = np.random.default_rng(2)
    for r, k, c in ((5, 7, 3), (64, 300, 33), (1, 1, 1)):
        A, B = random_query(rng, r * k).reshape(r, k), random_query(rng, k * c).reshape(k, c)
        want = (A.astype(np.uint64)[:, :, None] * B.astype(np.uint64)[None, :, :] & 0xFFFFFFFF).sum(axis=1) & 0xFFFFFFFF
        assert np.array_equal(orc.mul(A, B), want.astype(np.uint32))


def test_matrix_dimension_errors(orc):
    """matrix.rs:1251-1273 test_cases"""
    z = np.zeros
    assert orc.mul(z((1024, 1), np.uint32), z((1, 1024), np.uint32)).shape == (1024, 1024)
    with pytest.raises(orc.OracleError) as e:
        orc.mul(z((1024, 1), np.uint32), z((1024, 1), np.uint32))
    assert e.value.code == orc.ERR_INCOMPATIBLE_DIM_MATMUL
    with pytest.raises(orc.OracleError) as e:
        orc.add(z((1024, 1), np.uint32), z((1, 1024), np.uint32))
    assert e.value.code == orc.ERR_INCOMPATIBLE_DIM_MATADD


def test_matrix_addition_is_correct(orc):
    """matrix.rs:1378-1418: A + (-A) = 0"""
    rng = np.random.default_rng(3)
    A = orc.generate_from_seed(37, 91, rng.bytes(32))
    neg = (0 - A.astype(np.int64)).astype(np.uint32)
    assert not orc.add(A, neg).any()


def test_row_vector_compressed_transposed_matrix_multiplication_works(orc):
    """matrix.rs:1319-1376: all-ones matrix, random dims and bit length -> every output = wrapping sum of the vector"""
    rng = np.random.default_rng(4)
    for _ in range(60):
        N, C, b = int(rng.integers(1, 1025)), int(rng.integers(1, 1025)), int(rng.integers(4, 15))
        q = orc.generate_from_seed(1, N, rng.bytes(32))
        dtc = orc.row_wise_compress(orc.transpose(np.ones((N, C), dtype=np.uint32)), b)
        r = orc.row_vector_x_compressed_transposed_matrix(q, dtc, N, b)
        assert r.shape == (1, C) and np.array_equal(r[0], np.full(C, q.sum(dtype=np.uint32), dtype=np.uint32))


@pytest.mark.parametrize("b", ALL_BITS)
def test_respond_two_ways(b, orc):
    """the packed mat-vec equals the plain q*D product (mod 2^32) for every bit length and every N mod cf"""
    rng = np.random.default_rng(50 + b)
    cf = cf_of(b)
    for tail in range(cf):
        N, C = cf * 40 + tail, 11
        D = random_db_matrix(rng, N, C, b)
        q = random_query(rng, N)
        dtc = orc.row_wise_compress(orc.transpose(D), b)
        assert dtc.shape == (C, -(-N // cf))
        assert np.array_equal(orc.row_vector_x_compressed_transposed_matrix(q, dtc, N, b), orc.mul(q.reshape(1, -1), D))


def test_respond_dimension_errors(orc):
    """matrix.rs:329-331"""
    dtc = np.zeros((4, 3), dtype=np.uint32)
    for q in (np.zeros((1, 8), np.uint32), np.zeros((9, 1), np.uint32), np.zeros((2, 9), np.uint32)):
        with pytest.raises(orc.OracleError) as e:
            orc.row_vector_x_compressed_transposed_matrix(q, dtc, 9, 9)
        assert e.value.code == orc.ERR_INCOMPATIBLE_DIM_ROWVEC_X_TRANSPOSED


def test_compress_decompress_round_trip_and_bit_length_errors(orc):
    """matrix.rs:1520-1604 (on random masked matrices) and matrix.rs:99-101"""
    rng = np.random.default_rng(5)
    for b in ALL_BITS:
        r, c = int(rng.integers(1, 200)), int(rng.integers(1, 400))
        M = random_db_matrix(rng, r, c, b)
        assert np.array_equal(orc.row_wise_decompress(orc.row_wise_compress(M, b), b, c), M)
    for b in (0, 3, 15, 16):
        with pytest.raises(orc.OracleError) as e:
            orc.row_wise_compress(np.zeros((2, 2), np.uint32), b)
        assert e.value.code == orc.ERR_IMPOSSIBLE_ELEMENT_BIT_LENGTH


def test_compress_masks_and_zero_fills_tail(orc):
    """matrix.rs:121-125: fields are masked to b bits; missing tail fields are zero"""
    M = np.array([[0xFFFFFFFF] * 4], dtype=np.uint32)
    assert list(orc.row_wise_compress(M, 9)[0]) == [0x1FF | (0x1FF << 10) | (0x1FF << 20), 0x1FF]
    assert list(orc.row_wise_compress(M, 12)[0]) == [0xFFF | (0xFFF << 16)] * 2
    assert list(orc.row_wise_compress(M, 4)[0]) == [0x0F0F0F0F]


def test_serialized_matrix_can_be_deserialized(orc):
    """matrix.rs:1448-1486 and the from_bytes failure modes (matrix.rs:978-999)"""
    rng = np.random.default_rng(6)
    for _ in range(20):
        r, c = int(rng.integers(1, 300)), int(rng.integers(1, 300))
        A = orc.generate_from_seed(r, c, rng.bytes(32))
        img = orc.matrix_to_bytes(A)
        assert img == wire(A)
        assert np.array_equal(orc.matrix_from_bytes(img), A)
    good = wire(np.arange(6, dtype=np.uint32).reshape(2, 3))
    for bad in (b"", good[:8], good[:-1], good + b"\0", wire(np.zeros((0, 3), np.uint32)) + b"\0" * 4):
        with pytest.raises(orc.OracleError) as e:
            orc.matrix_from_bytes(bad)
        assert e.value.code == orc.ERR_FAILED_TO_DESERIALIZE_MATRIX


# ---- encoder: BFF + row codec ------------------------------------------------------------------------------------------
def _random_kv(rng, n, max_val=64, min_key=16):
    keys = {}
    while len(keys) < n:
        keys[rng.bytes(int(rng.integers(min_key, 33)))] = None
    return list(keys), [rng.bytes(int(rng.integers(1, max_val + 1))) for _ in keys]


def test_encode_kv_as_row_and_recover(orc):
    """serialization.rs:244-314: key 1..32 B x value 1..64 B x b 7..11 x all admissible row lengths"""
    rng = np.random.default_rng(7)
    for key_len in (1, 5, 16, 32):
        for val_len in (1, 2, 7, 33, 64):
            for b in range(7, 12):
                key, val = rng.bytes(key_len), rng.bytes(val_len)
                hashed = orc.turboshake128(key, 32)
                lo = -(-(32 * 8 + (val_len + 1) * 8) // b)
                hi = -(-(32 * 8 + (2 * val_len + 1) * 8) // b)
                for cols in range(lo, hi):
                    row = orc.encode_kv_as_row(key, val, b, cols)
                    assert int(row.max()) < (1 << b)
                    dec = orc.decode_kv_from_row(row, b)
                    assert dec[:32] == hashed and dec[32:] == val


@pytest.mark.parametrize("arity", [3, 4])
def test_encode_kv_database_and_recover_values(arity, orc):
    """matrix.rs:1136-1232: DB -> D -> recover every value, random sizes and bit lengths"""
    rng = np.random.default_rng(80 + arity)
    for n in (1, 2, 3, 256, 1000, 4096):
        b = int(rng.integers(4, 15))
        keys, vals = _random_kv(rng, n, max_val=80)
        D, filt, used = orc.from_kv_database(arity, keys, vals, b, rng.bytes(3200))
        assert D.shape == (filt.num_fingerprints, orc.encoded_num_cols(max(map(len, vals)), b))
        assert int(D.max()) < (1 << b) and filt.filter_size == n and filt.arity == arity
        for k, v in zip(keys, vals):
            assert orc.recover_value(D, filt, k) == v
        f2 = orc.Filter.from_bytes(filt.to_bytes())  # 68-byte round trip, binary_fuse_filter.rs:462-513
        assert f2 == filt


def test_empty_database_and_bad_filter_bytes(orc):
    with pytest.raises(orc.OracleError) as e:
        orc.from_kv_database(3, [], [], 8, bytes(3200))
    assert e.value.code == orc.ERR_EMPTY_KV_DATABASE  # matrix.rs:1430-1446
    with pytest.raises(orc.OracleError) as e:
        orc.Filter.from_bytes(bytes(67))
    assert e.value.code == orc.ERR_FAILED_TO_DESERIALIZE_FILTER


def test_bits_per_entry(orc):
    """validate_bits_per_entry_* (matrix.rs:1488-1518) depends only on the filter shape: N*b/n <= ceil(1.13 b) / ceil(1.08 b)"""
    for arity, factor in ((3, 1.13), (4, 1.08)):
        _, _, nf = orc.bff_shape(arity, 1_000_000)
        assert nf * 10 / 1_000_000 <= np.ceil(10 * factor)


# ---- end-to-end keyword PIR on the oracle alone (test_pir.rs:12-142) ---------------------------------------------------
@pytest.mark.parametrize("arity", [3, 4])
def test_keyword_pir_oracle_end_to_end(arity, orc):
    rng = np.random.default_rng(90 + arity)
    n = 700
    keys, vals = _random_kv(rng, n)
    seed = rng.bytes(32)
    b = orc.find_encoded_db_matrix_element_bit_length(n)
    D, filt, _ = orc.from_kv_database(arity, keys, vals, b, rng.bytes(3200))
    hint, dtc = orc.server_setup_from_matrix(seed, D, b)
    N = filt.num_fingerprints
    A = orc.generate_from_seed(1774, N, seed)
    ok = 0
    for key, val in list(zip(keys, vals))[:40]:
        s, e = orc.ternary_vector(1774, rng), orc.ternary_vector(N, rng)
        try:
            qb, sc = orc.client_query(A, hint, filt, key, s, e)
        except orc.OracleError as err:
            assert err.code == orc.ERR_ARITHMETIC_OVERFLOW_ADDING_QUERY_INDICATOR
            continue
        resp = unwire(orc.server_respond(dtc, N, b, wire(qb)))
        assert orc.client_process_response(filt, key, sc, resp) == val
        ok += 1
    assert ok >= 10
    # a key that is not in the database does not decode to a value (hash check fails)
    s, e = orc.ternary_vector(1774, rng), orc.ternary_vector(N, rng)
    qb, sc = orc.client_query(A, hint, filt, b"definitely-not-a-key", s, e)
    resp = unwire(orc.server_respond(dtc, N, b, wire(qb)))
    with pytest.raises(orc.OracleError):
        orc.client_process_response(filt, b"definitely-not-a-key", sc, resp)


def test_vectorised_ternary_sampler_equals_the_scalar_restatement(orc):
    """oracle.ternary_vector_np (used for the N-long error vectors of the full-size end-to-end test) draws the same u32 stream and maps
    it like or_ternary_from_u32 (matrix.rs:577-612), rejections included"""
    a = orc.ternary_vector(5000, np.random.default_rng(11))
    b = orc.ternary_vector_np(5000, np.random.default_rng(11))
    assert np.array_equal(a, b) and set(np.unique(a).tolist()) == {0, 1, 0xFFFFFFFF}
    # the rejection branch itself: values just around the boundaries
    import ctypes as C

    interval = (0xFFFFFFFF - 2) // 3
    for v, want in ((0, 0), (interval, 0), (interval + 1, 1), (2 * interval, 1), (2 * interval + 1, 0xFFFFFFFF), (3 * interval, 0xFFFFFFFF),
                    (3 * interval + 1, None), (0xFFFFFFFF, None)):
        t = C.c_uint32()
        ok = orc.lib().or_ternary_from_u32(C.c_uint32(v), C.byref(t))
        assert (t.value if ok else None) == want


def test_filter_slots_are_where_the_query_indicator_lands(orc):
    """oracle.filter_slots restates the slot derivation of Client::query (client.rs:109-113): a query for a key differs from s*A + e in
    exactly those slots, by 2^32 / 2^b"""
    rng = np.random.default_rng(21)
    for arity in (3, 4):
        keys = [rng.bytes(20) for _ in range(300)]
        vals = [rng.bytes(10) for _ in keys]
        b = orc.find_encoded_db_matrix_element_bit_length(len(keys))
        D, filt, _ = orc.from_kv_database(arity, keys, vals, b, rng.bytes(32 * 100))
        N = filt.num_fingerprints
        seed = rng.bytes(32)
        A = orc.generate_from_seed(1774, N, seed)
        hint = orc.mul(A, D)
        for key in keys[:5]:
            s, e = orc.ternary_vector(1774, rng), orc.ternary_vector_np(N, rng)
            try:
                q, _ = orc.client_query(A, hint, filt, key, s, e)
            except orc.OracleError:
                continue
            base = (orc.mul(s.reshape(1, -1), A)[0] + e).astype(np.uint32)
            diff = (q - base).astype(np.uint32)
            slots = orc.filter_slots(filt, key)
            assert len(set(slots)) == arity and sorted(np.nonzero(diff)[0].tolist()) == sorted(slots)
            assert all(int(diff[h]) == orc.query_indicator(b) for h in slots)
