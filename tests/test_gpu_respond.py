"""GPU parity: the HIP respond path (through the C ABI) vs the CPU oracle, bit-exact.

Restates row_vector_compressed_transposed_matrix_multiplication_works (reference matrix.rs:1319-1376) and adds what the
reference cannot test (it has no GPU respond): every element bit length, every N mod cf, ragged row counts, the wire
format and its error behaviour (matrix.rs:973-1010, 329-331), re-entrancy, shards and batches."""
import os
import threading

import numpy as np
import pytest

from _cases import ALL_BITS, OwnMapping, cf_of, random_db_matrix, random_query, unwire, wire

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["reference-packing", "dense64-where-offered", "planar-where-offered"])
def packing(request, native):
    """every test of this module runs once with the reference packing forced and once with the default (dense64 for
    b in {7, 9, 11, 12}); results must be bit-identical either way"""
    import chalametpir_amd as cp

    cp.tuning_set("layout.dense", 0 if request.param.startswith("reference") else 1)
    cp.tuning_set("layout.planar", 1 if request.param.startswith("planar") else 0)
    yield request.param
    cp.tuning_set("layout.dense", 1)
    cp.tuning_set("layout.planar", 1)


def make_server(cp, orc, device, rng, N, C, b):
    D = random_db_matrix(rng, N, C, b)
    dtc = orc.row_wise_compress(orc.transpose(D), b)  # server.rs:64-67 on the oracle
    srv = cp.Server.from_compressed(dtc, N, b, device=device)
    return srv, dtc


@pytest.mark.parametrize("b", ALL_BITS)
def test_respond_matches_oracle_every_bit_length_and_tail(b, orc, device):
    import chalametpir_amd as cp

    rng = np.random.default_rng(1000 + b)
    cf = cf_of(b)
    for tail in range(cf):  # N mod cf covers every tail-word case (matrix.rs:360-375, 399-421, 446-475)
        for base, C in ((cf * 5, 3), (cf * 1024 * 2, 17), (cf * 1500, 40)):
            N = base + tail
            srv, dtc = make_server(cp, orc, device, rng, N, C, b)
            for _ in range(2):
                q = random_query(rng, N)
                want = orc.row_vector_x_compressed_transposed_matrix(q, dtc, N, b)[0]
                got = srv.respond_array(q)
                assert np.array_equal(got, want), (b, N, C)


def test_all_ones_matrix_gives_sum_of_query(orc, device):
    """reference property test matrix.rs:1319-1376: all-ones DB -> every output = wrapping sum of q"""
    import chalametpir_amd as cp

    rng = np.random.default_rng(7)
    for _ in range(12):
        N, C, b = int(rng.integers(1, 1025)), int(rng.integers(1, 1025)), int(rng.integers(4, 15))
        ones = np.ones((N, C), dtype=np.uint32)
        dtc = orc.row_wise_compress(orc.transpose(ones), b)
        srv = cp.Server.from_compressed(dtc, N, b, device=device)
        q = orc.generate_from_seed(1, N, rng.bytes(32))[0]
        got = srv.respond_array(q)
        assert np.array_equal(got, np.full(C, q.sum(dtype=np.uint32), dtype=np.uint32))


def test_single_slot_and_single_column(orc, device):
    import chalametpir_amd as cp

    rng = np.random.default_rng(3)
    for N, C, b in ((1, 1, 4), (1, 5, 9), (2, 1, 14), (3, 1, 10)):
        srv, dtc = make_server(cp, orc, device, rng, N, C, b)
        q = random_query(rng, N)
        assert np.array_equal(srv.respond_array(q), orc.row_vector_x_compressed_transposed_matrix(q, dtc, N, b)[0])


def test_import_ignores_garbage_beyond_n_and_above_b_bits(orc, device):
    """the reference masks every field with 2^b-1 and bounds-checks the query index in the last word
    (matrix.rs:352-357, 360-375); a compressed matrix carrying garbage there must give the same answer"""
    import chalametpir_amd as cp

    rng = np.random.default_rng(11)
    for b in (4, 9, 12):
        cf = cf_of(b)
        N, C = cf * 700 + 1, 9
        D = random_db_matrix(rng, N, C, b)
        dtc = orc.row_wise_compress(orc.transpose(D), b)
        dirty = dtc.copy()
        slot = 32 // cf
        high = (((1 << slot) - 1) ^ ((1 << b) - 1)) if slot > b else 0
        junk = 0
        for j in range(cf):
            junk |= high << (j * slot)
        dirty |= np.uint32(junk & 0xFFFFFFFF)
        dirty[:, -1] |= np.uint32(0xFFFFFFFF << slot & 0xFFFFFFFF)  # fields beyond N in the last word
        q = random_query(rng, N)
        want = orc.row_vector_x_compressed_transposed_matrix(q, dirty, N, b)[0]
        assert np.array_equal(want, orc.row_vector_x_compressed_transposed_matrix(q, dtc, N, b)[0])
        srv = cp.Server.from_compressed(dirty, N, b, device=device)
        assert np.array_equal(srv.respond_array(q), want)


def test_respond_wire_format_and_errors(orc, device):
    """Server::respond on bytes (server.rs:184-190): same bytes as the oracle, same error variants as the reference"""
    import chalametpir_amd as cp

    rng = np.random.default_rng(5)
    N, C, b = 3001, 23, 10
    srv, dtc = make_server(cp, orc, device, rng, N, C, b)
    q = random_query(rng, N)
    query = wire(q)
    got = srv.respond(query)
    assert got == orc.server_respond(dtc, N, b, query)
    assert len(got) == 8 + 4 * C and unwire(got).shape == (1, C)

    def variant(fn):
        with pytest.raises(cp.ChalametPIRError) as e:
            fn()
        return e.value.variant

    # Matrix::from_bytes failures (matrix.rs:978-999)
    assert variant(lambda: srv.respond(b"")) == "FailedToDeserializeMatrixFromBytes"
    assert variant(lambda: srv.respond(query[:8])) == "FailedToDeserializeMatrixFromBytes"
    assert variant(lambda: srv.respond(query[:-1])) == "FailedToDeserializeMatrixFromBytes"
    assert variant(lambda: srv.respond(query + b"\0\0\0\0")) == "FailedToDeserializeMatrixFromBytes"
    assert variant(lambda: srv.respond(wire(np.zeros((0, 5), dtype=np.uint32)) + b"\0" * 4)) == "FailedToDeserializeMatrixFromBytes"
    # dimension failures (matrix.rs:329-331)
    assert variant(lambda: srv.respond(wire(random_query(rng, N - 1)))) == "IncompatibleDimensionForRowVectorTransposedMatrixMultiplication"
    assert variant(lambda: srv.respond(wire(q.reshape(-1, 1)))) == "IncompatibleDimensionForRowVectorTransposedMatrixMultiplication"
    # the same inputs produce the same codes on the oracle
    for bad in (b"", query[:8], query[:-1], wire(random_query(rng, N - 1)), wire(q.reshape(-1, 1))):
        with pytest.raises(orc.OracleError) as oe:
            orc.server_respond(dtc, N, b, bad)
        with pytest.raises(cp.ChalametPIRError) as pe:
            srv.respond(bad)
        assert oe.value.code == pe.value.code


def test_invalid_bit_length_rejected(device):
    import chalametpir_amd as cp

    for b in (0, 3, 15, 16, 32):
        with pytest.raises(cp.ChalametPIRError) as e:
            cp.Server.from_compressed(np.zeros((2, 2), dtype=np.uint32), 4, b, device=device)
        assert e.value.variant == "ImpossibleEncodedDBMatrixElementBitLength"  # matrix.rs:99-101


def test_respond_is_reentrant_and_clone_shares_db(orc, device):
    """respond(&self) is called concurrently on an Arc<Server> (reference examples/server.rs:45,55,85)"""
    import chalametpir_amd as cp

    rng = np.random.default_rng(9)
    N, C, b = 3 * 4096 + 2, 64, 9
    srv, dtc = make_server(cp, orc, device, rng, N, C, b)
    clone = srv.clone()
    queries = [random_query(rng, N) for _ in range(24)]
    wants = [orc.row_vector_x_compressed_transposed_matrix(q, dtc, N, b)[0] for q in queries]
    results = [None] * len(queries)

    def work(i):
        s = srv if i % 2 == 0 else clone
        for _ in range(3):
            results[i] = s.respond_array(queries[i])

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(queries))]
    [t.start() for t in threads]
    [t.join() for t in threads]
    for got, want in zip(results, wants):
        assert np.array_equal(got, want)
    srv.close()
    assert np.array_equal(clone.respond_array(queries[0]), wants[0])  # the clone keeps the DB alive


def test_every_kernel_variant_is_bit_identical(orc, device):
    import chalametpir_amd as cp

    rng = np.random.default_rng(21)
    N, C, b = 3 * 1024 * 20 + 1, 100, 9
    srv, dtc = make_server(cp, orc, device, rng, N, C, b)
    q = random_query(rng, N)
    want = orc.row_vector_x_compressed_transposed_matrix(q, dtc, N, b)[0]
    try:
        for nt in (0, 1):
            for xs in (0, 1):
                for bpc in (0, 1, 3):
                    cp.tuning_set("respond.nontemporal", nt)
                    cp.tuning_set("respond.xcd_split", xs)
                    cp.tuning_set("respond.blocks_per_cu", bpc)
                    assert np.array_equal(srv.respond_array(q), want), (nt, xs, bpc)
    finally:
        cp.tuning_set("respond.nontemporal", 1)
        cp.tuning_set("respond.xcd_split", 1)
        cp.tuning_set("respond.blocks_per_cu", 2)


def test_device_entry_points_shards_and_batches(orc, device):
    """device-pointer ABI used by the multi-GPU path: shard partials sum (wrap-around) to the full response, and a batch
    equals independent responds"""
    import torch

    import chalametpir_amd as cp
    from chalametpir_amd.distributed import shard_range

    rng = np.random.default_rng(31)
    for b in (9, 12, 6):
        cf = cf_of(b)
        N, C = cf * 1024 * 9 + cf * 100 + 1, 37
        D = random_db_matrix(rng, N, C, b)
        dtc_full = orc.row_wise_compress(orc.transpose(D), b)
        q = random_query(rng, N)
        want = orc.row_vector_x_compressed_transposed_matrix(q, dtc_full, N, b)[0]
        q_dev = torch.from_numpy(q.view(np.int32)).cuda()
        total = np.zeros(C, dtype=np.uint32)
        world = 3
        unit = cp.dtc_layout_for(N, C, b).slots_per_chunk
        for rank in range(world):
            lo, hi = shard_range(N, unit, rank, world)
            if hi <= lo:
                continue
            D_dev = torch.from_numpy(np.ascontiguousarray(D[lo:hi]).view(np.int32)).cuda()
            srv = cp.Server.from_device_matrix(D_dev, hi - lo, C, b, device=device, slot_offset=lo, total_slots=N)
            r_dev = torch.empty(C, dtype=torch.int32, device="cuda")
            srv.respond_device(q_dev, r_dev, stream=torch.cuda.current_stream())
            torch.cuda.synchronize()
            total += r_dev.cpu().numpy().view(np.uint32)
        assert np.array_equal(total, want), b

        # batch of 11 on the unsharded server: fused (2 passes of 4, one of 2, one of 1) and unfused (11 passes in one launch),
        # each in both pass orders
        srv = cp.Server.from_compressed(dtc_full, N, b, device=device)
        Q = np.stack([random_query(rng, N) for _ in range(11)])
        want_all = [orc.row_vector_x_compressed_transposed_matrix(Q[i], dtc_full, N, b)[0] for i in range(11)]
        Q_dev = torch.from_numpy(Q.view(np.int32)).cuda()
        try:
            for fusion in (1, 0):
                for order in (0, 1):
                    cp.tuning_set("respond.batch_fusion", fusion)
                    cp.tuning_set("respond.interleave_passes", order)
                    R_dev = torch.full((11, C), -1, dtype=torch.int32, device="cuda")
                    srv.respond_batch_device(Q_dev, 11, R_dev, stream=torch.cuda.current_stream())
                    torch.cuda.synchronize()
                    got = R_dev.cpu().numpy().view(np.uint32)
                    for i in range(11):
                        assert np.array_equal(got[i], want_all[i]), (b, fusion, order, i)
        finally:
            cp.tuning_set("respond.batch_fusion", 1)
            cp.tuning_set("respond.interleave_passes", -1)


def test_unaligned_queries_odd_shard_offsets_and_every_batch_size(orc, device):
    """the guarded / scalar query paths of the kernels: a query that is only 4-byte aligned, a query length that is not a multiple
    of 4 with several queries per launch, a shard that starts at an odd slot, and every batch size 1..17 (fused and unfused)"""
    import torch

    import chalametpir_amd as cp

    rng = np.random.default_rng(77)
    stream = torch.cuda.current_stream()
    for b in (9, 10, 13, 6):
        cf = cf_of(b)
        N, C = 3 * 1024 * cf + 7, 21  # N % 4 == 3 for every cf
        D = random_db_matrix(rng, N, C, b)
        dtc = orc.row_wise_compress(orc.transpose(D), b)
        srv = cp.Server.from_compressed(dtc, N, b, device=device)
        nq = 17
        Q = np.stack([random_query(rng, N) for _ in range(nq)])
        want = np.stack([orc.row_vector_x_compressed_transposed_matrix(Q[i], dtc, N, b)[0] for i in range(nq)])
        Q_dev = torch.from_numpy(Q.view(np.int32)).cuda()
        lo, hi = 1027, N - 5
        D_dev = torch.from_numpy(np.ascontiguousarray(D[lo:hi]).view(np.int32)).cuda()
        shard = cp.Server.from_device_matrix(D_dev, hi - lo, C, b, device=device, slot_offset=lo, total_slots=N)
        dtc_shard = orc.row_wise_compress(orc.transpose(D[lo:hi]), b)
        want_part = orc.row_vector_x_compressed_transposed_matrix(Q[0][lo:hi], dtc_shard, hi - lo, b)[0]
        try:
            # respond.ks_major: 1 = the wide kernel for every launch, 2 = the step-major kernel wherever it applies (passes of up to 4)
            for ks_major in (1, 2):
                cp.tuning_set("respond.ks_major", ks_major)
                # one query, buffer shifted by one element: 4-byte but not 16-byte aligned
                buf = torch.zeros(N + 8, dtype=torch.int32, device="cuda")
                for shift in (1, 2, 3):
                    buf[shift:shift + N] = torch.from_numpy(Q[0].view(np.int32)).cuda()
                    r = torch.full((C,), -1, dtype=torch.int32, device="cuda")
                    srv.respond_device(buf[shift:shift + N], r, stream=stream)
                    torch.cuda.synchronize()
                    assert np.array_equal(r.cpu().numpy().view(np.uint32), want[0]), (b, ks_major, shift)
                # every batch size; rows of Q_dev are N apart and N % 4 != 0, so all but the first are unaligned
                for fusion in (1, 0):
                    cp.tuning_set("respond.batch_fusion", fusion)
                    for k in range(1, nq + 1):
                        R = torch.full((k, C), -1, dtype=torch.int32, device="cuda")
                        srv.respond_batch_device(Q_dev, k, R, stream=stream)
                        torch.cuda.synchronize()
                        assert np.array_equal(R.cpu().numpy().view(np.uint32), want[:k]), (b, ks_major, fusion, k)
                cp.tuning_set("respond.batch_fusion", 1)
                # a shard that starts at an odd slot and ends in the middle of a packing unit
                part = torch.empty(C, dtype=torch.int32, device="cuda")
                shard.respond_device(Q_dev[0], part, stream=stream)
                torch.cuda.synchronize()
                assert np.array_equal(part.cpu().numpy().view(np.uint32), want_part), (b, ks_major)
        finally:
            cp.tuning_set("respond.batch_fusion", 1)
            cp.tuning_set("respond.ks_major", 1)


def test_query_in_page_locked_memory_takes_the_direct_upload(orc, device):
    """cpir_host_alloc: a query in page-locked memory is uploaded straight from the caller's buffer; same answers"""
    import chalametpir_amd as cp

    rng = np.random.default_rng(123)
    N, C, b = 5 * 1536 + 3, 19, 9
    srv, dtc = make_server(cp, orc, device, rng, N, C, b)
    pin = cp.PinnedArray(N)
    for _ in range(3):
        q = random_query(rng, N)
        pin.array[:] = q
        want = orc.row_vector_x_compressed_transposed_matrix(q, dtc, N, b)[0]
        assert np.array_equal(srv.respond_array(pin.array), want)
        assert np.array_equal(srv.respond_array(q), want)
    pin.close()


def test_lone_host_query_is_read_in_place(orc, device):
    """a caller that finds the server idle is served without an upload: the step-major kernel reads its query in place -- from the
    caller's buffer when that is page-locked and 16-byte aligned, else from the server's pinned block, filled and launched in two
    halves once the query has 2^19 words.  Same answers as with respond.host_zero_copy=0 (upload first), for a whole server and for a
    shard that starts in the middle of the query."""
    import torch

    import chalametpir_amd as cp

    rng = np.random.default_rng(4242)
    for b, N, C in ((9, 3 * (1 << 18) + 77, 7), (6, (1 << 19) + 2048 + 5, 5), (10, 5 * 1536 + 3, 19)):
        D = random_db_matrix(rng, N, C, b)
        dtc = orc.row_wise_compress(orc.transpose(D), b)
        whole = cp.Server.from_compressed(dtc, N, b, device=device)
        lo, hi = 1024 * 3, N - 1000  # a shard: reads only q[lo:hi], from where it lies
        D_dev = torch.from_numpy(np.ascontiguousarray(D[lo:hi]).view(np.int32)).cuda()
        shard = cp.Server.from_device_matrix(D_dev, hi - lo, C, b, device=device, slot_offset=lo, total_slots=N)
        dtc_shard = orc.row_wise_compress(orc.transpose(D[lo:hi]), b)
        pin = cp.PinnedArray(N + 4)
        try:
            for trial in range(2):
                q = random_query(rng, N)
                want = orc.row_vector_x_compressed_transposed_matrix(q, dtc, N, b)[0]
                want_shard = orc.row_vector_x_compressed_transposed_matrix(q[lo:hi], dtc_shard, hi - lo, b)[0]
                for zero_copy in (1, 0):
                    cp.tuning_set("respond.host_zero_copy", zero_copy)
                    for shift in (0, 1):  # page-locked and aligned: read in place; shifted by one word: staged like pageable memory
                        view = pin.array[shift:shift + N]
                        view[:] = q
                        assert np.array_equal(whole.respond_array(view), want), (b, zero_copy, shift)
                        assert np.array_equal(shard.respond_array(view), want_shard), (b, zero_copy, shift)
                    assert np.array_equal(whole.respond_array(q), want), (b, zero_copy)  # pageable
                    assert np.array_equal(shard.respond_array(q), want_shard), (b, zero_copy)
        finally:
            cp.tuning_set("respond.host_zero_copy", 1)
            pin.close()


def test_a_few_concurrent_callers_share_one_pass_that_reads_their_queries_in_place(orc, device, packing):
    """respond.inplace_seats (default 4): two to four concurrent callers are answered by ONE pass of the step-major kernel that reads every
    query over the host link where it lies -- a page-locked query in its caller's buffer (a table of row addresses), a pageable one in the
    server's pinned block WHILE its caller copies it in (every seat's progress polled) -- no upload.  Same answers as the oracle for a whole
    server and for a shard that starts in the middle of the query, for page-locked, pageable and mixed rounds, beside a caller whose
    page-locked view is not 16-byte aligned (copied like a pageable one); six callers are too many for such rounds (upload path); with the
    key at 0 no such round is ever formed; with a 1 us limit on a query long enough for the pass to catch up with its copy the polled passes
    give up, are answered again, and after three of them pageable callers go back to the upload path."""
    import torch

    import chalametpir_amd as cp

    rng = np.random.default_rng(20251)
    b, N, C = 9, (1 << 19) + 5 * 512 + 37, 21
    D = random_db_matrix(rng, N, C, b)
    dtc = orc.row_wise_compress(orc.transpose(D), b)
    lo, hi = 1024 * 5, N - 777
    dtc_shard = orc.row_wise_compress(orc.transpose(D[lo:hi]), b)
    n_q = 6
    qs = [random_query(rng, N) for _ in range(n_q)]
    want = [orc.row_vector_x_compressed_transposed_matrix(q, dtc, N, b)[0] for q in qs]
    want_shard = [orc.row_vector_x_compressed_transposed_matrix(q[lo:hi], dtc_shard, hi - lo, b)[0] for q in qs]
    pins = [cp.PinnedArray(N + 4) for _ in range(n_q)]
    for pa, q in zip(pins, qs):
        pa.array[:N] = q
    shifted = cp.PinnedArray(N + 4)
    shifted.array[1:N + 1] = qs[0]
    # ... and a caller whose own mapping (never heap memory: see OwnMapping) is hipHostRegister'ed: page-locked by registration
    rt = torch.cuda.cudart()
    own = OwnMapping(N) if hasattr(rt, "cudaHostRegister") and hasattr(rt, "cudaHostUnregister") else None
    if own is not None:
        own.array[:] = qs[1]
        assert int(rt.cudaHostRegister(own.address, own.nbytes, 0)) == 0
    planar = packing.startswith("planar")  # (the other packings have no kernel that reads a query exactly once)
    # who asks: "p" a page-locked query, "g" a pageable one, "s" the page-locked view that is not 16-byte aligned, "r" the registered mapping
    crews = ("pp", "ppp", "pppp", "gg", "pg", "ggg", "pgsg", "ppggps") + (("rp", "rgr") if own is not None else ())
    try:
        for seats, timeout_us in ((4, 2000), (2, 2000), (0, 2000)):
            cp.tuning_set("respond.inplace_seats", seats)
            cp.tuning_set("respond.host_fill_timeout_us", timeout_us)
            whole = cp.Server.from_compressed(dtc, N, b, device=device)
            D_dev = torch.from_numpy(np.ascontiguousarray(D[lo:hi]).view(np.int32)).cuda()
            shard = cp.Server.from_device_matrix(D_dev, hi - lo, C, b, device=device, slot_offset=lo, total_slots=N)
            for srv, wanted in ((whole, want), (shard, want_shard)):
                calls = 0
                for crew in crews:
                    bad = []
                    start = threading.Barrier(len(crew))

                    def ask(t, kind):
                        start.wait()
                        for i in range(10):
                            k = (t + i) % n_q
                            if kind == "g":
                                got = srv.respond_array(qs[k])
                            elif kind == "s":
                                got, k = srv.respond_array(shifted.array[1:N + 1]), 0
                            elif kind == "r":
                                got, k = srv.respond_array(own.array), 1
                            else:
                                got = srv.respond_array(pins[k].array[:N])
                            if not np.array_equal(got, wanted[k]):
                                bad.append((t, i, k))

                    ts = [threading.Thread(target=ask, args=(t, kind)) for t, kind in enumerate(crew)]
                    for t in ts:
                        t.start()
                    for t in ts:
                        t.join()
                    assert not bad, (seats, timeout_us, crew, bad[:4])
                    calls += 10 * len(crew)
                counts = srv.host_path_counts()
                assert counts["calls"] == calls, counts
                assert counts["calls"] == counts["alone"] + counts["in_uploaded_rounds"] + counts["in_in_place_rounds"], counts
                if seats == 0 or not planar:
                    assert counts["in_place_rounds"] == 0 and counts["in_in_place_rounds"] == 0, counts
                else:
                    assert counts["in_place_rounds"] > 0 and counts["in_in_place_rounds"] <= seats * counts["in_place_rounds"], counts
                    assert counts["polled_passes_given_up"] == 0, counts
            whole.close()
            shard.close()
        # a query long enough for the pass to catch up with its copy (16 MB: a millisecond of one thread's memcpy), and a 1 us limit
        if planar:
            N2, C2 = (1 << 22) + 1024, 4
            D2 = random_db_matrix(rng, N2, C2, b)
            dtc2 = orc.row_wise_compress(orc.transpose(D2), b)
            q2 = [random_query(rng, N2) for _ in range(2)]
            want2 = [orc.row_vector_x_compressed_transposed_matrix(q, dtc2, N2, b)[0] for q in q2]
            cp.tuning_set("respond.inplace_seats", 4)
            cp.tuning_set("respond.host_fill_timeout_us", 1)
            srv = cp.Server.from_compressed(dtc2, N2, b, device=device)
            bad = []
            start = threading.Barrier(2)

            def ask2(t):
                start.wait()
                for i in range(8):
                    if not np.array_equal(srv.respond_array(q2[(t + i) % 2]), want2[(t + i) % 2]):
                        bad.append((t, i))

            ts = [threading.Thread(target=ask2, args=(t,)) for t in range(2)]
            for t in ts:
                t.start()
            for t in ts:
                t.join()
            counts = srv.host_path_counts()
            assert not bad, (bad, counts)
            assert counts["in_place_rounds"] > 0 and counts["polled_passes_given_up"] >= 1, counts  # (three in a row: no more polling)
            srv.close()
    finally:
        cp.tuning_set("respond.inplace_seats", 4)
        cp.tuning_set("respond.host_fill_timeout_us", 2000)
        for pa in pins:
            pa.close()
        shifted.close()
        if own is not None:
            err = rt.cudaHostUnregister(own.address)
            own.close()
            assert int(err) == 0, err


def test_partly_registered_query_buffer_is_not_read_in_place(orc, device):
    """hipHostRegister over only the first half of a query buffer: the in-place path must see that the END of the range is not
    page-locked and stage the query instead (a kernel reading unmapped host pages would fault); registered as a whole it is read in place.
    Same answers either way."""
    import torch

    import chalametpir_amd as cp

    rt = torch.cuda.cudart()
    if not hasattr(rt, "cudaHostRegister") or not hasattr(rt, "cudaHostUnregister"):
        pytest.skip("no hipHostRegister binding in this torch")
    rng = np.random.default_rng(99)
    N, C, b = 6 * 4096 + 512, 11, 9
    srv, dtc = make_server(cp, orc, device, rng, N, C, b)
    own = OwnMapping(N)  # a mapping of its own, never heap memory (see OwnMapping)
    q = own.array
    q[:] = random_query(rng, N)
    want = orc.row_vector_x_compressed_transposed_matrix(q, dtc, N, b)[0]
    half_bytes = (N // 2 * 4) // 4096 * 4096
    for nbytes in (half_bytes, own.nbytes):
        err = rt.cudaHostRegister(own.address, nbytes, 0)
        assert int(err) == 0, err
        try:
            for _ in range(2):
                assert np.array_equal(srv.respond_array(q), want), nbytes
        finally:
            err = rt.cudaHostUnregister(own.address)
        assert int(err) == 0, err
        # unregistered again: the same address must not be taken for page-locked any more
        assert np.array_equal(srv.respond_array(q), want)
    srv.close()
    del q
    own.close()


def test_lone_pageable_query_polled_launch_and_its_fallbacks(orc, device):
    """a lone pageable query of 2^19+ words: ONE launch in front of the copy, the kernel waiting for each step's words
    (respond.host_fill_timeout_us, default 2 ms per wave and step).  With a 1 us limit every wave gives up at once: the launch is flagged void and
    the query answered again from the complete pinned block; after three such launches the server stops polling (two launches, each
    when its half is in place); 0 switches polling off from the start.  Same answers throughout."""
    import chalametpir_amd as cp

    rng = np.random.default_rng(777)
    b, N, C = 9, (1 << 19) + 3 * 4096 + 17, 6
    D = random_db_matrix(rng, N, C, b)
    dtc = orc.row_wise_compress(orc.transpose(D), b)
    qs = [random_query(rng, N) for _ in range(3)]
    want = [orc.row_vector_x_compressed_transposed_matrix(q, dtc, N, b)[0] for q in qs]
    try:
        for timeout_us in (20000, 1, 0):
            cp.tuning_set("respond.host_fill_timeout_us", timeout_us)
            srv = cp.Server.from_compressed(dtc, N, b, device=device)  # a fresh server: its give-up count starts at zero
            for rep in range(3):  # with the 1 us limit: three void launches, then the two-launch path
                for q, w in zip(qs, want):
                    assert np.array_equal(srv.respond_array(q), w), (timeout_us, rep)
            srv.close()
    finally:
        cp.tuning_set("respond.host_fill_timeout_us", 2000)


def test_random_shapes_every_kernel_order(orc, device):
    """seeded random shapes (slots, columns, bit length, shard window, query alignment) through every dispatch of the matrix-core
    respond: the wide kernel (1, the default), the step-major kernel (2) and the step-major kernel in the strided step order of the in-place
    host path (3) -- block counts above and below the number of steps, ragged last steps, shards that start at odd slots -- against
    the oracle; the host entry point on top (lone caller: query read in place / staged)."""
    import torch

    import chalametpir_amd as cp

    rng = np.random.default_rng(20260)
    stream = torch.cuda.current_stream()
    try:
        for trial in range(40):
            b = int(rng.choice([4, 6, 8, 9, 9, 10, 12, 14]))
            cf = cf_of(b)
            steps = int(rng.choice([1, 2, 3, 7, 31, 200, 257, 300, 700]))  # around 256 blocks of the one-block-per-CU grid
            N = max(cf, steps * 512 - int(rng.integers(0, 512)))
            C = int(rng.choice([1, 3, 16, 17, 63, 64, 65, 130]))
            if N * C > 12_000_000:
                C = max(1, 12_000_000 // N)
            D = random_db_matrix(rng, N, C, b)
            lo = int(rng.integers(0, max(1, N // 3)))
            hi = N - int(rng.integers(0, max(1, N // 5)))
            if rng.integers(0, 3) == 0:
                lo, hi = 0, N
            dtc = orc.row_wise_compress(orc.transpose(D[lo:hi]), b)
            D_dev = torch.from_numpy(np.ascontiguousarray(D[lo:hi]).view(np.int32)).cuda()
            srv = cp.Server.from_device_matrix(D_dev, hi - lo, C, b, device=device, slot_offset=lo, total_slots=N)
            q = random_query(rng, N)
            want = orc.row_vector_x_compressed_transposed_matrix(q[lo:hi], dtc, hi - lo, b)[0]
            shift = int(rng.integers(0, 4))
            buf = torch.zeros(N + 8, dtype=torch.int32, device="cuda")
            buf[shift:shift + N] = torch.from_numpy(q.view(np.int32)).cuda()
            planar = srv.layout.packing == 2
            for mode in (1, 2, 3):
                if mode == 3 and not planar:
                    continue
                cp.tuning_set("respond.ks_major", mode)
                r = torch.full((C,), -1, dtype=torch.int32, device="cuda")
                srv.respond_device(buf[shift:shift + N], r, stream=stream)
                torch.cuda.synchronize()
                assert np.array_equal(r.cpu().numpy().view(np.uint32), want), (trial, b, N, C, lo, hi, shift, mode)
            cp.tuning_set("respond.ks_major", 1)
            assert np.array_equal(srv.respond_array(q), want), (trial, "host")
            srv.close()
    finally:
        cp.tuning_set("respond.ks_major", 1)


def test_wide_database_fused_batches_go_window_by_window(orc, device):
    """more columns than the kernels' LDS accumulators hold for a fused pass (step-major kernel, 4 queries: 3 072 columns; wide kernel, 19
    queries: about 1 700): the launch is repeated over column windows; every batch size around the window arithmetic gives the same
    responses as single responds, wide and step-major dispatch alike"""
    import torch

    import chalametpir_amd as cp

    rng = np.random.default_rng(515)
    stream = torch.cuda.current_stream()
    for b, N, C in ((9, 2 * 512 + 77, 3100), (6, 700, 1601)):
        D = random_db_matrix(rng, N, C, b)
        dtc = orc.row_wise_compress(orc.transpose(D), b)
        srv = cp.Server.from_compressed(dtc, N, b, device=device)
        nq = 19
        Q = np.stack([random_query(rng, N) for _ in range(nq)])
        want = np.stack([orc.row_vector_x_compressed_transposed_matrix(Q[i], dtc, N, b)[0] for i in range(nq)])
        Q_dev = torch.from_numpy(Q.view(np.int32)).cuda()
        for ks_major in (1, 2):
            cp.tuning_set("respond.ks_major", ks_major)
            for k in (1, 2, 3, 4, 5, 8, 9, 12, 13, 16, 19):  # (9..12: one pass on three row sets; 13+: a pass of 12 and the rest)
                R = torch.full((k, C), -1, dtype=torch.int32, device="cuda")
                srv.respond_batch_device(Q_dev, k, R, stream=stream)
                torch.cuda.synchronize()
                assert np.array_equal(R.cpu().numpy().view(np.uint32), want[:k]), (b, ks_major, k)
        cp.tuning_set("respond.ks_major", 1)
        assert np.array_equal(srv.respond_array(Q[0]), want[0])  # the host entry point (one query: one window)
        srv.close()


def test_more_columns_than_one_query_fits_in_the_accumulators(orc, device):
    """beyond 12 288 columns even ONE query's responses exceed the step-major kernel's 48 KiB of LDS accumulators: a lone device launch goes
    over two column windows, the host entry point stages the query instead of reading it in place (a query behind the host link must be
    read once), and everything agrees with the wide kernel and the oracle"""
    import torch

    import chalametpir_amd as cp

    rng = np.random.default_rng(12400)
    stream = torch.cuda.current_stream()
    b, N, C = 9, 600, 12400
    D = random_db_matrix(rng, N, C, b)
    dtc = orc.row_wise_compress(orc.transpose(D), b)
    srv = cp.Server.from_compressed(dtc, N, b, device=device)
    Q = np.stack([random_query(rng, N) for _ in range(3)])
    want = np.stack([orc.row_vector_x_compressed_transposed_matrix(Q[i], dtc, N, b)[0] for i in range(3)])
    Q_dev = torch.from_numpy(Q.view(np.int32)).cuda()
    for ks_major in (1, 2):
        cp.tuning_set("respond.ks_major", ks_major)
        r = torch.full((C,), -1, dtype=torch.int32, device="cuda")
        srv.respond_device(Q_dev[0], r, stream=stream)
        R = torch.full((3, C), -1, dtype=torch.int32, device="cuda")
        srv.respond_batch_device(Q_dev, 3, R, stream=stream)
        torch.cuda.synchronize()
        assert np.array_equal(r.cpu().numpy().view(np.uint32), want[0]), ks_major
        assert np.array_equal(R.cpu().numpy().view(np.uint32), want), ks_major
    cp.tuning_set("respond.ks_major", 1)
    pin = cp.PinnedArray(N)
    pin.array[:] = Q[1]
    assert np.array_equal(srv.respond_array(Q[1]), want[1]) and np.array_equal(srv.respond_array(pin.array), want[1])
    pin.close()
    srv.close()


def test_in_place_rounds_take_only_as_many_callers_as_the_accumulators_hold(orc, device, packing):
    """an in-place round (respond.inplace_seats) is ONE pass of the step-major kernel, whose 48 KiB of LDS accumulators hold one u32 per
    query and padded column: at 5 000 columns two queries fit (rounds of two, whatever the key allows), at 7 312 -- 8 kB values -- one:
    no such rounds, four concurrent page-locked callers go through the upload path.  Same answers either way."""
    import chalametpir_amd as cp

    rng = np.random.default_rng(7312)
    b, N = 9, 5 * 512 + 100
    for C, fits in ((5000, 2), (7312, 1)):
        D = random_db_matrix(rng, N, C, b)
        dtc = orc.row_wise_compress(orc.transpose(D), b)
        srv = cp.Server.from_compressed(dtc, N, b, device=device)
        qs = [random_query(rng, N) for _ in range(4)]
        want = [orc.row_vector_x_compressed_transposed_matrix(q, dtc, N, b)[0] for q in qs]
        pins = [cp.PinnedArray(N) for _ in range(4)]
        for pa, q in zip(pins, qs):
            pa.array[:] = q
        bad = []
        start = threading.Barrier(4)

        def ask(t):
            start.wait()
            for i in range(10):
                k = (t + i) % 4
                if not np.array_equal(srv.respond_array(pins[k].array), want[k]):
                    bad.append((t, i))

        ts = [threading.Thread(target=ask, args=(t,)) for t in range(4)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        counts = srv.host_path_counts()
        assert not bad, (C, bad[:4], counts)
        assert counts["calls"] == 40 and counts["in_in_place_rounds"] <= fits * counts["in_place_rounds"], (C, counts)
        if fits == 1 or not packing.startswith("planar"):
            assert counts["in_place_rounds"] == 0, (C, counts)
        for pa in pins:
            pa.close()
        srv.close()


def test_environment_cannot_change_a_response(orc, device):
    """Server::respond has no mode in which it lies (server.rs:184-190).  Rounds 3-4 had measuring aids in the release library that the
    environment could switch on (CPIR_WIDE_ABLATE skipped parts of the wide kernel, CPIR_KS_TRACE made launches synchronous): they now
    exist only in the -DCPIR_DIAG build.  A process with every such variable set answers exactly as the oracle does -- device queries,
    fused batches, a lone host query -- and the tuning key of the matmul ablation is refused."""
    import hashlib
    import subprocess
    import sys

    rng = np.random.default_rng(31415)
    b, N, C = 9, 5 * 512 + 77, 45
    D = random_db_matrix(rng, N, C, b)
    dtc = orc.row_wise_compress(orc.transpose(D), b)
    Q = np.stack([random_query(rng, N) for _ in range(30)])
    want = np.stack([orc.row_vector_x_compressed_transposed_matrix(q, dtc, N, b)[0] for q in Q])
    np.savez("/tmp/cpir_env_case.npz", dtc=dtc, Q=Q, N=N, b=b)
    code = (
        "import sys, hashlib, numpy as np, torch\n"
        f"sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})\n"
        "import chalametpir_amd as cp\n"
        "z = np.load('/tmp/cpir_env_case.npz'); dtc, Q, N, b = z['dtc'], z['Q'], int(z['N']), int(z['b'])\n"
        "srv = cp.Server.from_compressed(dtc, N, b, device=cp.Device(0))\n"
        "s = torch.cuda.current_stream(); Qd = torch.from_numpy(Q.view(np.int32)).cuda()\n"
        "R = torch.empty((30, dtc.shape[0]), dtype=torch.int32, device='cuda'); srv.respond_batch_device(Qd, 30, R, stream=s)\n"
        "r1 = torch.empty(dtc.shape[0], dtype=torch.int32, device='cuda'); srv.respond_device(Qd[3], r1, stream=s); torch.cuda.synchronize()\n"
        "h = srv.respond_array(Q[5])\n"
        "try:\n    cp.tuning_set('matmul.ablate', 1); refused = False\nexcept cp.ChalametPIRError:\n    refused = True\n"
        "print(hashlib.sha256(R.cpu().numpy().tobytes() + r1.cpu().numpy().tobytes() + h.tobytes()).hexdigest(), refused)\n"
    )
    expected = hashlib.sha256(want.astype(np.uint32).tobytes() + want[3].astype(np.uint32).tobytes() + want[5].astype(np.uint32).tobytes()).hexdigest()
    for extra in ({}, {"CPIR_WIDE_ABLATE": "7", "CPIR_KS_TRACE": "1", "CPIR_MM_ABLATE": "15"}):
        p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-3000:]
        digest, refused = p.stdout.split()[-2:]
        assert digest == expected and refused == "True", (extra, p.stdout[-300:])
        assert "[ks trace]" not in p.stderr and "[wide trace]" not in p.stderr
