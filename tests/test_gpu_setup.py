"""GPU parity of the offline path: transpose+compress, the hint matmul, Server::setup from a matrix and from a KV database,
and the end-to-end "client decodes the right value" property (reference integrations/src/test_pir.rs:12-142)."""
import numpy as np
import pytest

from _cases import ALL_BITS, cf_of, random_db_matrix, random_query, unwire, wire

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["reference-packing", "dense64-where-offered", "planar-where-offered"])
def packing(request, native):
    import chalametpir_amd as cp

    cp.tuning_set("layout.dense", 0 if request.param.startswith("reference") else 1)
    cp.tuning_set("layout.planar", 1 if request.param.startswith("planar") else 0)
    yield request.param
    cp.tuning_set("layout.dense", 1)
    cp.tuning_set("layout.planar", 1)


def _dev(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint32).view(np.int32)).cuda()


def _host(t):
    return t.cpu().numpy().view(np.uint32)


@pytest.mark.parametrize("b", ALL_BITS)
def test_transpose_compress_matches_oracle(b, orc, device):
    """gpu_utils::mat_transpose + Matrix::row_wise_compress (reference gpu_utils.rs:222-281, matrix.rs:98-205, 517-527)"""
    import torch

    import chalametpir_amd as cp

    rng = np.random.default_rng(200 + b)
    cf = cf_of(b)
    # (the last two shapes are wide enough for the whole-rows pack kernel -- 512+ columns, whole and ragged steps, C % 4 != 0)
    for N, C in ((cf * 64 + 1, 5), (cf * 300 + cf - 1, 66), (1, 1), (cf * 2048, 130), (777, 64), (1536 + 77, 530), (512, 1031)):
        D = random_db_matrix(rng, N, C, 16)  # entries wider than b: compress must mask them (matrix.rs:121)
        want = orc.row_wise_compress(orc.transpose(D), b)
        srv = cp.Server.from_device_matrix(_dev(D), N, C, b, device=device)
        assert np.array_equal(srv.export_compressed(), want), (b, N, C)
        # padding of the device layout is zero and the OR flag reports the widest entry
        L = srv.layout
        dtc = torch.empty(L.total_words, dtype=torch.int32, device="cuda")
        flag = torch.zeros(1, dtype=torch.int32, device="cuda")
        device.transpose_compress(_dev(D), L, dtc, or_of_entries=flag, stream=torch.cuda.current_stream())
        torch.cuda.synchronize()
        assert int(_host(flag)[0]) == int(np.bitwise_or.reduce(D, axis=None))
        if L.packing == 2:  # planar: low bytes XOR 0x80 + bit planes in MFMA operand order, then the column sums
            check_planar_image(_host(dtc), L, D, b)
            # both pack kernels (64-column waves / whole rows per block), whatever the width: the same image, the same OR
            for rows_mode in (0, 1):
                cp.tuning_set("pack.rows", rows_mode)
                other = torch.full((L.total_words,), -1, dtype=torch.int32, device="cuda")
                flag.zero_()
                device.transpose_compress(_dev(D), L, other, or_of_entries=flag, stream=torch.cuda.current_stream())
                torch.cuda.synchronize()
                assert torch.equal(other, dtc), (b, N, C, rows_mode)
                assert int(_host(flag)[0]) == int(np.bitwise_or.reduce(D, axis=None))
            cp.tuning_set("pack.rows", -1)
            continue
        img = _host(dtc).reshape(L.rows_padded, L.words_per_row_padded)
        assert not img[C:].any()  # padded rows are zero
        if L.packing == 0:  # reference packing: the device image IS the reference matrix plus zero padding
            assert np.array_equal(img[:C, : L.words_per_row], want)
            assert not img[:, L.words_per_row:].any()
        else:  # dense64: K fields of b bits per u64; every field lives at the documented (chunk, plane, position)
            K = L.fields_per_word
            img64 = img.view(np.uint64)
            n = np.arange(N, dtype=np.uint64)
            chunk, within = n // (K * 1024), n % (K * 1024)
            j, p = within // 1024, within % 1024
            m = ((p >> 1) & 1) * 512 + 2 * (p >> 2) + (p & 1)
            got = (img64[:C][:, (chunk * 1024 + m).astype(np.int64)] >> (j * b).astype(np.uint64)) & np.uint64((1 << b) - 1)
            assert np.array_equal(got.astype(np.uint32), (D & ((1 << b) - 1)).T)
            assert int(np.count_nonzero(img64[:C])) <= N * C  # nothing but fields: unused positions and top bits stay zero


def check_planar_image(words, L, D, b):
    """decode the planar device image (layout documented at cpir_dtc_layout in include/chalamet_hip.h) with numpy, independently of
    the library's own export kernel: every (slot, column) incl. padding, and the per-column field sums behind the tiles"""
    N, C = D.shape
    hb = max(b - 8, 0)  # b <= 8: the byte alone
    ks_total = -(-N // 512)
    tile_bytes = (8 + hb) * 1024
    assert (L.chunk_words, L.slots_per_chunk, L.words_per_row_padded) == (tile_bytes // 4, 512, ks_total * (8 + hb) * 16)
    assert L.total_words == L.rows_padded * L.words_per_row_padded + L.rows_padded
    n_tiles_words = L.rows_padded * L.words_per_row_padded
    raw = words[:n_tiles_words].view(np.uint8)
    n = np.arange(ks_total * 512, dtype=np.int64)[:, None]
    c = np.arange(L.rows_padded, dtype=np.int64)[None, :]
    T, cl = c >> 4, c & 15
    ks, s = n >> 9, n & 511
    kb, g, j = s >> 6, (s >> 4) & 3, s & 15
    lane = g * 16 + cl
    tile0 = (T * ks_total + ks) * tile_bytes
    f = (raw[tile0 + kb * 1024 + lane * 16 + j] ^ 0x80).astype(np.uint32)
    for p in range(hb):
        off = tile0 + 8192 + p * 1024 + lane * 16 + (kb >> 1) * 4
        word = sum(raw[off + i].astype(np.uint32) << (8 * i) for i in range(4))
        f |= ((word >> (8 * (j & 3) + 4 * (kb & 1) + (j >> 2)).astype(np.uint32)) & 1) << (8 + p)
    want = np.zeros((ks_total * 512, L.rows_padded), dtype=np.uint32)
    want[:N, :C] = D & ((1 << b) - 1)
    assert np.array_equal(f, want)  # fields where they belong, zero fields in every padding slot and padding column
    assert np.array_equal(words[n_tiles_words:], want.sum(axis=0, dtype=np.uint64).astype(np.uint32))


def test_compress_then_decompress_round_trip(orc, device):
    """row_wise_compressed_matrix_can_be_decompressed (reference matrix.rs:1520-1604), with the GPU doing the compress"""
    import chalametpir_amd as cp

    rng = np.random.default_rng(77)
    for b in (5, 9, 10, 13):
        N, C = int(rng.integers(1000, 5000)), int(rng.integers(10, 200))
        D = random_db_matrix(rng, N, C, b)
        srv = cp.Server.from_device_matrix(_dev(D), N, C, b, device=device)
        back = orc.row_wise_decompress(srv.export_compressed(), b, N)
        assert np.array_equal(back, orc.transpose(D))


def test_mat_x_mat_identity_property(orc, device):
    """matrix_multiplication_is_correct (reference matrix.rs:1275-1317): A*I = A and I*A = A for random dims 1..1024"""
    import torch

    rng = np.random.default_rng(41)
    seed = rng.bytes(32)
    for _ in range(10):
        rows, cols = int(rng.integers(1, 1025)), int(rng.integers(1, 1025))
        A = orc.generate_from_seed(rows, cols, seed)
        for lhs, rhs, bits in ((A, orc.identity(cols), 16), (A, orc.identity(cols), 32), (orc.identity(rows), A, 32)):
            M = torch.empty((lhs.shape[0], rhs.shape[1]), dtype=torch.int32, device="cuda")
            device.mat_x_mat(_dev(lhs), _dev(rhs), M, lhs.shape[0], lhs.shape[1], rhs.shape[1], rhs_max_bits=bits,
                             stream=torch.cuda.current_stream())
            torch.cuda.synchronize()
            assert np.array_equal(_host(M), A), (rows, cols, bits)


def test_mat_x_mat_matches_oracle_random(orc, device):
    """impl Mul for &Matrix (reference matrix.rs:1040-1059) on random operands: packed-16 and general kernels, split-K,
    ragged edges, accumulate mode"""
    import torch

    rng = np.random.default_rng(43)
    for rows, inner, cols, bits in ((1774, 2000, 940, 9), (64, 70001, 128, 14), (65, 33, 129, 16), (3, 5000, 7, 32),
                                    (200, 4096, 300, 32), (1, 1, 1, 16), (130, 100000, 20, 10)):
        A = random_query(rng, rows * inner).reshape(rows, inner)
        D = rng.integers(0, 1 << bits, size=(inner, cols), dtype=np.uint64).astype(np.uint32)
        want = orc.mul(A, D)
        M = torch.empty((rows, cols), dtype=torch.int32, device="cuda")
        device.mat_x_mat(_dev(A), _dev(D), M, rows, inner, cols, rhs_max_bits=bits, stream=torch.cuda.current_stream())
        torch.cuda.synchronize()
        assert np.array_equal(_host(M), want), (rows, inner, cols, bits)
        # accumulate: K split in two halves must add up to the same product (this is how N-shards combine)
        h = inner // 2
        if h:
            M.zero_()
            device.mat_x_mat(_dev(A[:, :h]), _dev(D[:h]), M, rows, h, cols, rhs_max_bits=bits, accumulate=True)
            device.mat_x_mat(_dev(A[:, h:]), _dev(D[h:]), M, rows, inner - h, cols, rhs_max_bits=bits, accumulate=True)
            device.synchronize()
            torch.cuda.synchronize()
            assert np.array_equal(_host(M), want), ("accumulate", rows, inner, cols, bits)


def test_products_back_to_back_on_one_stream_keep_their_scratch_to_themselves(orc, device):
    """three hint products enqueued back to back on ONE stream (the shards of a small group: inner 3072 / 3072 / 3584, 20 columns), 300
    times over: each keeps its prepared right-hand side in scratch memory of its own until its kernels are done.  (With the runtime's
    stream-ordered allocator the second and third product read a recycled block before its kernel had written it -- one iteration in a
    hundred wrong in every entry, scripts/probes/matmul_backtoback_repro.cpp, mallocasync_order_probe.hip; the library allocates such
    scratch itself now and frees it behind an event.)"""
    import torch

    rng = np.random.default_rng(3072)
    rows, cols, b = 1774, 20, 9
    inners = (3072, 3072, 3584)
    stream = torch.cuda.current_stream()
    ops = []
    for inner in inners:
        A = random_query(rng, rows * inner).reshape(rows, inner)
        D = rng.integers(0, 1 << b, size=(inner, cols), dtype=np.uint64).astype(np.uint32)
        D[rng.random(inner) < 0.2] = 0
        ops.append((_dev(A), _dev(D), torch.empty((rows, cols), dtype=torch.int32, device="cuda"), orc.mul(A, D), inner))
    for it in range(300):
        for A_dev, D_dev, M, _, inner in ops:
            M.fill_(-1)
            device.mat_x_mat(A_dev, D_dev, M, rows, inner, cols, rhs_max_bits=16, stream=stream)
        torch.cuda.synchronize()
        for g, (_, _, M, want, inner) in enumerate(ops):
            assert np.array_equal(_host(M), want), (it, g, inner)


def test_mat_x_mat_on_the_matrix_cores(orc, device):
    """the i8 matrix-core kernel (csrc/matmul_mfma.hip) against the oracle's impl Mul (matrix.rs:1040-1059) and against the VALU kernel:
    ragged rows / columns / k tails, padded leading dimensions, extreme byte patterns in both operands (every limb 0x00, 0x7f, 0x80,
    0xff), full 16-bit right-hand sides, accumulate mode"""
    import torch

    import chalametpir_amd as cp

    assert cp.mat_x_mat_kernel_name(16) == "mat_x_mat_mfma_pipe_kernel" and cp.mat_x_mat_kernel_name(32) == "mat_x_mat_u32_kernel"
    rng = np.random.default_rng(47)
    stream = torch.cuda.current_stream()
    extremes_a = np.array([0, 0xFFFFFFFF, 0x80808080, 0x7F7F7F7F, 0x00FF807F, 0x80000000, 1, 0x01010101], dtype=np.uint32)
    extremes_d = np.array([0, 0xFFFF, 0x8080, 0x7F7F, 0x00FF, 0xFF00, 0x0080, 0x8000, 1], dtype=np.uint32)
    for rows, inner, cols, pad in ((1, 4, 1, 0), (128, 64, 128, 0), (129, 64 * 5 + 4, 17, 4), (257, 8192 + 60, 940, 8), (1774, 64 * 9, 130, 0),
                                   (300, 4 * 33333, 33, 12), (16, 1 << 18, 16, 0)):
        A = random_query(rng, rows * inner).reshape(rows, inner)
        D = rng.integers(0, 1 << 16, size=(inner, cols), dtype=np.uint64).astype(np.uint32)
        A.reshape(-1)[rng.integers(0, A.size, size=min(A.size, 4096))] = rng.choice(extremes_a, size=min(A.size, 4096))
        D.reshape(-1)[rng.integers(0, D.size, size=min(D.size, 4096))] = rng.choice(extremes_d, size=min(D.size, 4096))
        if rows >= 3:
            A[1] = 0xFFFFFFFF
            A[2] = 0x80808080
        if cols >= 3:
            D[:, 1] = 0xFFFF
            D[:, 2] = 0
        want = orc.mul(A, D)
        lda, ldd, ldm = inner + pad, cols + pad, cols + (pad // 4)
        A_dev = torch.full((rows, lda), -1, dtype=torch.int32, device="cuda")
        D_dev = torch.full((inner, ldd), -1, dtype=torch.int32, device="cuda")
        A_dev[:, :inner] = _dev(A)
        D_dev[:, :cols] = _dev(D)
        got = {}
        try:
            for mfma in (1, 0):
                cp.tuning_set("matmul.mfma", mfma)
                M = torch.full((rows, ldm), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
                device.mat_x_mat(A_dev, D_dev, M, rows, inner, cols, lda=lda, ldd=ldd, ldm=ldm, rhs_max_bits=16, stream=stream)
                torch.cuda.synchronize()
                got[mfma] = _host(M)
                assert np.array_equal(got[mfma][:, :cols], want), (rows, inner, cols, mfma)
                assert np.all(got[mfma][:, cols:] == 0x5A5A5A5A)  # padding of M untouched
        finally:
            cp.tuning_set("matmul.mfma", 1)
        # accumulate: the K axis in two unequal parts (multiples of 4) adds up to the product on top of what M held
        h = (inner // 3) // 4 * 4
        if h:
            base = random_query(rng, rows * cols).reshape(rows, cols)
            M = _dev(base).clone()
            device.mat_x_mat(A_dev[:, :h].contiguous(), D_dev[:h], M, rows, h, cols, ldd=ldd, rhs_max_bits=16, accumulate=True, stream=stream)
            device.mat_x_mat(A_dev[:, h:inner].contiguous(), D_dev[h:], M, rows, inner - h, cols, ldd=ldd, rhs_max_bits=16, accumulate=True,
                             stream=stream)
            torch.cuda.synchronize()
            assert np.array_equal(_host(M), base + want), ("accumulate", rows, inner, cols)


def test_mat_x_packed_takes_the_packed_image_as_right_hand_side(orc, device):
    """cpir_op_transpose_compress_with_plane + cpir_op_mat_x_packed == impl Mul for &Matrix (matrix.rs:1040-1059) on the unpacked D:
    the low-byte operand pieces of the planar image + the high-byte plane written in the same pass are the right-hand side of the
    matrix-core matmul.  One to six bit planes, ragged rows / columns / k tails and super-tiles, extreme limbs in A, accumulate mode; the
    image is bit-identical to the one cpir_op_transpose_compress writes; not offered for b <= 8 or the other packings."""
    import torch

    import chalametpir_amd as cp

    rng = np.random.default_rng(4711)
    stream = torch.cuda.current_stream()
    extremes_a = np.array([0, 0xFFFFFFFF, 0x80808080, 0x7F7F7F7F, 0x00FF807F, 0x80000000, 1, 0x01010101], dtype=np.uint32)
    for b, rows, N, C in ((9, 1, 4, 1), (9, 128, 512, 16), (10, 129, 64 * 5 + 4, 17), (9, 257, 8192 + 60, 940), (12, 300, 4 * 3333, 33),
                          (14, 77, 512 * 9 - 4, 130), (11, 1774, 64 * 17, 20), (9, 16, 1 << 17, 16)):
        L = cp.dtc_layout_for(N, C, b, packing=2)
        # one bit plane (b = 9): the matmul expands the high byte from the image's own plane, no plane is written; more: one byte per field
        assert L.packing == 2 and cp.packed_rhs_offered(L)
        assert cp.packed_rhs_plane_bytes(L) == (0 if b == 9 else (L.rows_padded // 16) * ((N + 63) // 64) * 1024)
        A = random_query(rng, rows * N).reshape(rows, N)
        A.reshape(-1)[rng.integers(0, A.size, size=min(A.size, 4096))] = rng.choice(extremes_a, size=min(A.size, 4096))
        if rows >= 3:
            A[1] = 0xFFFFFFFF
            A[2] = 0x80808080
        D = random_db_matrix(rng, N, C, b)
        if C >= 3:
            D[:, 1] = (1 << b) - 1
            D[:, 2] = 0
        want = orc.mul(A, D)
        A_dev, D_dev = _dev(A), _dev(D)
        dtc = torch.full((L.total_words,), -1, dtype=torch.int32, device="cuda")
        dtc_plain = torch.full((L.total_words,), -1, dtype=torch.int32, device="cuda")
        plane = torch.full((cp.packed_rhs_plane_bytes(L) // 4,), -1, dtype=torch.int32, device="cuda") if cp.packed_rhs_plane_bytes(L) else None
        flag = torch.zeros(1, dtype=torch.int32, device="cuda")
        device.transpose_compress_with_plane(D_dev, L, dtc, plane, or_of_entries=flag, stream=stream)
        device.transpose_compress(D_dev, L, dtc_plain, stream=stream)
        torch.cuda.synchronize()
        assert torch.equal(dtc, dtc_plain)
        for rows_mode in (0, 1):  # both pack kernels write the same image and the same high-byte plane
            cp.tuning_set("pack.rows", rows_mode)
            dtc2 = torch.full_like(dtc, -1)
            plane2 = torch.full_like(plane, -1) if plane is not None else None
            device.transpose_compress_with_plane(D_dev, L, dtc2, plane2, stream=stream)
            torch.cuda.synchronize()
            assert torch.equal(dtc2, dtc) and (plane is None or torch.equal(plane2, plane)), (b, N, C, rows_mode)
        cp.tuning_set("pack.rows", -1)
        assert int(flag.item()) >> b == 0
        ldm = C + 3
        M = torch.full((rows, ldm), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
        device.mat_x_packed(A_dev, dtc, L, plane, M, rows, ldm=ldm, stream=stream)
        torch.cuda.synchronize()
        got = _host(M)
        assert np.array_equal(got[:, :C], want), (b, rows, N, C)
        assert np.all(got[:, C:] == 0x5A5A5A5A)  # padding of M untouched
        base = random_query(rng, rows * C).reshape(rows, C)
        M2 = _dev(base).clone()
        device.mat_x_packed(A_dev, dtc, L, plane, M2, rows, accumulate=True, stream=stream)
        torch.cuda.synchronize()
        assert np.array_equal(_host(M2), base + want), ("accumulate", b, rows, N, C)
    # a plane where none belongs (b = 9) and none where one does (b = 10) are refused
    some = torch.zeros(1 << 16, dtype=torch.int32, device="cuda")
    for bb, pl in ((9, some), (10, None)):
        L = cp.dtc_layout_for(4096, 16, bb, packing=2)
        dtc = torch.zeros(L.total_words, dtype=torch.int32, device="cuda")
        with pytest.raises(cp.ChalametPIRError):
            device.transpose_compress_with_plane(some, L, dtc, pl, stream=stream)
        with pytest.raises(cp.ChalametPIRError):
            device.mat_x_packed(some, dtc, L, pl, some, 1, stream=stream)
    # where the pairing is not offered
    for L in (cp.dtc_layout_for(4096, 16, 8, packing=2), cp.dtc_layout_for(4096, 16, 9, packing=0)):
        assert cp.packed_rhs_plane_bytes(L) == 0 and not cp.packed_rhs_offered(L)
        dtc = torch.zeros(L.total_words, dtype=torch.int32, device="cuda")
        some = torch.zeros(1 << 16, dtype=torch.int32, device="cuda")
        with pytest.raises(cp.ChalametPIRError):
            device.transpose_compress_with_plane(some, L, dtc, some, stream=stream)
        with pytest.raises(cp.ChalametPIRError):
            device.mat_x_packed(some, dtc, L, some, some, 1, stream=stream)


def test_mat_x_mat_dimension_errors(device):
    import torch

    import chalametpir_amd as cp

    t = torch.zeros(16, dtype=torch.int32, device="cuda")
    with pytest.raises(cp.ChalametPIRError) as e:
        device.mat_x_mat(t, t, t, 0, 4, 4)
    assert e.value.variant == "InvalidMatrixDimension"  # Matrix::new, matrix.rs:45-55


# (the last case makes A 85 MB: two 64 MiB staging blocks, neither a multiple of the 168-byte sponge rate -- the squeeze continues
# across calls mid-block -- and two chunks of the pipelined hint matmul wait on different upload events)
# (N a multiple of 4 and b >= 9: the hint's right-hand side comes from the packed image itself + the high-byte plane written beside it --
# ragged last k-steps and super-tiles, one to six bit planes, column counts around the 16- and 128-column tiles; the other cases take the
# byte-plane split or the VALU kernels)
@pytest.mark.parametrize("b,N,C", [(9, 3 * 1100 + 1, 97), (10, 5000, 64), (12, 2049, 33), (7, 4097, 130), (9, 12001, 8),
                                   (9, 4 * 1537, 131), (14, 512 * 3, 17), (10, 516, 1), (11, 64 * 41, 129), (9, 512 * 8 + 60, 260)])
def test_setup_from_matrix_matches_oracle(b, N, C, orc, device):
    """Server::setup minus the encoder (reference server.rs:59-67): hint = A(seed)*D and the resident packed DB"""
    import chalametpir_amd as cp

    rng = np.random.default_rng(b * 1000 + N)
    seed = rng.bytes(32)
    D = random_db_matrix(rng, N, C, b)
    want_hint, want_dtc = orc.server_setup_from_matrix(seed, D, b)
    srv, hint = cp.Server.setup_from_matrix(seed, D, b, device=device)
    assert np.array_equal(hint, want_hint)
    assert np.array_equal(srv.export_compressed(), want_dtc)
    assert (srv.decompressed_num_cols, srv.mat_elem_bit_len, srv.response_len) == (N, b, C)
    q = random_query(rng, N)
    assert srv.respond(wire(q)) == orc.server_respond(want_dtc, N, b, wire(q))
    # caller-supplied A takes the same path as the seed-expanded A
    A = orc.generate_from_seed(1774, N, seed)
    srv2, hint2 = cp.Server.setup_from_matrix(seed, D, b, pub_mat_a=A, device=device)
    assert np.array_equal(hint2, want_hint)


def test_setup_uses_unmasked_entries_for_the_hint(orc, device):
    """the reference multiplies A by D before any masking (server.rs:61) but masks when compressing (matrix.rs:121):
    entries >= 2^16 must flip the hint product to the general kernel and still match"""
    import chalametpir_amd as cp

    rng = np.random.default_rng(99)
    N, C, b = 1500, 20, 9
    D = random_query(rng, N * C).reshape(N, C)  # full-range u32
    seed = rng.bytes(32)
    want_hint, want_dtc = orc.server_setup_from_matrix(seed, D, b)
    srv, hint = cp.Server.setup_from_matrix(seed, D, b, device=device)
    assert np.array_equal(hint, want_hint)
    assert np.array_equal(srv.export_compressed(), want_dtc)
    # entries that fit 16 bits but not b: the packed image (masked to b bits) must not serve as the hint's right-hand side
    N, C, b = 4 * 700, 37, 9
    D = rng.integers(0, 1 << 14, size=(N, C), dtype=np.uint64).astype(np.uint32)
    want_hint, want_dtc = orc.server_setup_from_matrix(seed, D, b)
    srv, hint = cp.Server.setup_from_matrix(seed, D, b, device=device)
    assert np.array_equal(hint, want_hint)
    assert np.array_equal(srv.export_compressed(), want_dtc)


def _random_kv(rng, n, max_val=64):
    keys = {}
    while len(keys) < n:
        keys[rng.bytes(int(rng.integers(16, 33)))] = None
    return {k: rng.bytes(int(rng.integers(1, max_val + 1))) for k in keys}


@pytest.mark.parametrize("arity", [3, 4])
def test_keyword_pir_end_to_end(arity, orc, device):
    """test_keyword_pir_with_{3,4}_wise_xor_filter (reference integrations/src/test_pir.rs:12-142): full Server::setup on a
    KV database -> client query -> Server::respond -> client decodes exactly the stored value.  The client is the
    oracle's restatement of chalametpir_client (client.rs:95-194, 209-275); the server is the HIP path."""
    import chalametpir_amd as cp

    rng = np.random.default_rng(500 + arity)
    for n in (300, 2000):
        db = _random_kv(rng, n)
        seed = rng.bytes(32)
        fseeds = rng.bytes(32 * 100)
        srv, hint_bytes, filter_bytes = cp.Server.setup(seed, db, arity, device=device, filter_seed_material=fseeds)
        # the same setup on the oracle gives the same bytes
        keys, vals = list(db.keys()), list(db.values())
        b = orc.find_encoded_db_matrix_element_bit_length(n)
        D, filt, _ = orc.from_kv_database(arity, keys, vals, b, fseeds)
        want_hint, want_dtc = orc.server_setup_from_matrix(seed, D, b)
        assert filter_bytes == filt.to_bytes() and len(filter_bytes) == 68
        assert hint_bytes == wire(want_hint)
        assert np.array_equal(srv.export_compressed(), want_dtc)
        # client side
        filt2 = orc.Filter.from_bytes(filter_bytes)
        hint = unwire(hint_bytes)
        assert hint.shape[0] == 1774  # Client::setup check, client.rs:47-49
        N = filt2.num_fingerprints
        A = orc.generate_from_seed(1774, N, seed)
        done = 0
        for key in keys[:200]:
            s = orc.ternary_vector(1774, rng)
            e = orc.ternary_vector(N, rng)
            try:
                qb, sc = orc.client_query(A, hint, filt2, key, s, e)
            except orc.OracleError as err:
                assert err.code == orc.ERR_ARITHMETIC_OVERFLOW_ADDING_QUERY_INDICATOR  # retried in the reference (test_pir.rs:66-70)
                continue
            resp = unwire(srv.respond(wire(qb)))
            assert orc.client_process_response(filt2, key, sc, resp) == db[key]
            done += 1
            if done == 10:
                break
        assert done == 10


def test_setup_error_behaviour(device):
    import chalametpir_amd as cp

    with pytest.raises(cp.ChalametPIRError) as e:
        cp.Server.setup(bytes(32), {}, 3, device=device)
    assert e.value.variant == "EmptyKVDatabase"  # server.rs:48-51
    with pytest.raises(cp.ChalametPIRError) as e:
        cp.Server.setup(bytes(32), {b"apple": b"red"}, 5, device=device)
    assert e.value.variant == "UnsupportedArityForBinaryFuseFilter"
    # a one-entry database is valid (reference test matrix.rs:1430-1446)
    srv, hint, fb = cp.Server.setup(bytes(32), {b"apple": b"red"}, 3, device=device)
    assert len(fb) == 68 and srv.mat_elem_bit_len == 14


def test_sharded_hint_partials_sum_to_the_hint(orc, device):
    """multi-GPU setup: hint = sum over N-shards of A[:, shard] * D[shard, :]; each shard's partial comes from
    cpir_hint_partial_device (A expanded from the seed on the host, only the shard's columns uploaded)"""
    import torch

    import chalametpir_amd as cp
    from chalametpir_amd.distributed import shard_range

    rng = np.random.default_rng(808)
    b, C = 9, 61
    N = 7 * 1024 * 2 + 3 * 1024 + 5
    seed = rng.bytes(32)
    D = random_db_matrix(rng, N, C, b)
    want = orc.mul(orc.generate_from_seed(1774, N, seed), D)
    unit = cp.dtc_layout_for(N, C, b).slots_per_chunk
    total = np.zeros((1774, C), dtype=np.uint32)
    world = 3
    for rank in range(world):
        lo, hi = shard_range(N, unit, rank, world)
        if hi <= lo:
            continue
        D_dev = _dev(D[lo:hi])
        M = torch.empty((1774, C), dtype=torch.int32, device="cuda")
        device.hint_partial(seed, D_dev, lo, hi - lo, N, C, M, stream=torch.cuda.current_stream())
        torch.cuda.synchronize()
        total += _host(M)
    assert np.array_equal(total, want)


def test_xof_on_this_host_matches_oracle(orc, native):
    """Matrix::generate_from_seed through the product's XOF on the GPU box's host CPU: there the x86-64-v3 clone of the Keccak
    permutation (chalametpir_amd/csrc/host_xof.cpp) is the one dispatched, which the build container's older CPU never runs.
    Marked gpu only because it has to run on that box; sizes straddle the 168-byte block boundary."""
    import chalametpir_amd as cp

    rng = np.random.default_rng(4242)
    for rows, cols in ((1, 1), (1, 41), (1, 42), (3, 335), (1, 336), (1, 337), (7, 100003), (1774, 3001)):
        seed = rng.bytes(32)
        got = cp.generate_from_seed(rows, cols, seed)
        assert got.tobytes() == orc.turboshake128(seed, rows * cols * 4), (rows, cols)
