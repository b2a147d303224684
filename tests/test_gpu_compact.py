"""Serving only the slots that hold something (chalametpir_amd/csrc/compact.hip).

A real encoded database leaves N - n of its N rows all zero (the filter slots no key owns, reference
chalametpir_common/src/matrix.rs:702-746); the server packs only the rows with a non-zero field and gathers every query onto
those slots in front of the kernel.  A zero row contributes 0 whatever the query holds there, so every response, the hint and the
exported compressed matrix must equal the oracle's on the FULL matrix bit for bit -- through every entry point: device queries
(single, batched fused / unfused), host queries (lone pageable / page-locked: compacted on the host while they are staged;
concurrent callers: gathered on the device), shards, the in-process group, import / export, and with the feature switched off."""
import threading

import time

import numpy as np
import pytest

from _cases import OwnMapping, cf_of, random_db_matrix, random_query, unwire, wire

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["reference-packing", "dense64-where-offered", "planar-where-offered"])
def packing(request, native):
    import chalametpir_amd as cp

    cp.tuning_set("layout.dense", 0 if request.param.startswith("reference") else 1)
    cp.tuning_set("layout.planar", 1 if request.param.startswith("planar") else 0)
    yield request.param
    cp.tuning_set("layout.dense", 1)
    cp.tuning_set("layout.planar", 1)


def holey_matrix(rng, N, C, b, zero_frac=0.12, pattern="random"):
    """an encoded-database look-alike: uniform fields, a share of the rows all zero"""
    D = random_db_matrix(rng, N, C, b)
    if pattern == "random":
        dead = rng.random(N) < zero_frac
    elif pattern == "runs":  # whole stretches, incl. the first and the last rows (a step of the kernel without a single kept slot)
        dead = np.zeros(N, dtype=bool)
        dead[: min(N, 700)] = True
        dead[-min(N, 300):] = True
        dead[N // 2: N // 2 + min(N // 4, 2000)] = True
    else:  # every second row
        dead = (np.arange(N) % 2) == 1
    if dead.all():
        dead[N // 3] = False
    D[dead] = 0
    return D, np.nonzero(~dead)[0].astype(np.uint32)


def dev(x):
    import torch

    return torch.from_numpy(np.ascontiguousarray(x).view(np.int32)).cuda()


def host(t):
    return t.cpu().numpy().view(np.uint32)


def oracle_dtc(orc, D, b):
    return orc.row_wise_compress(orc.transpose(D), b)


@pytest.mark.parametrize("b,pattern", [(9, "random"), (10, "runs"), (4, "random"), (7, "alternate"), (12, "runs"), (14, "random")])
def test_setup_from_matrix_compacts_and_answers_like_the_oracle(b, pattern, orc, device):
    import chalametpir_amd as cp

    rng = np.random.default_rng(9000 + b)
    cf = cf_of(b)
    for N, C in ((cf * 1024 * 3 + 5, 37), (5000, 64), (700, 3)):
        D, kept = holey_matrix(rng, N, C, b, pattern=pattern)
        seed = rng.bytes(32)
        srv, hint = cp.Server.setup_from_matrix(seed, D, b, device=device)
        want_hint, want_dtc = orc.server_setup_from_matrix(seed, D, b)
        served, of = srv.slots_served()
        assert (served, of) == (kept.size, N) and int(srv.physical_layout.num_slots) == kept.size and int(srv.layout.num_slots) == N
        assert np.array_equal(srv.kept_slots(), kept)
        assert np.array_equal(hint, want_hint)  # the hint is computed from ALL rows of D, unmasked (server.rs:61)
        assert np.array_equal(srv.export_compressed(), want_dtc)  # export spreads the kept slots over all N again
        for _ in range(3):
            q = random_query(rng, N)
            want = orc.row_vector_x_compressed_transposed_matrix(q, want_dtc, N, b)[0]
            assert np.array_equal(srv.respond_array(q), want)
            assert unwire(srv.respond(wire(q)))[0].tolist() == want.tolist()
        # wire errors are those of the logical database (matrix.rs:329-331)
        with pytest.raises(cp.ChalametPIRError) as e:
            srv.respond(wire(random_query(rng, kept.size)))
        assert e.value.variant == "IncompatibleDimensionForRowVectorTransposedMatrixMultiplication"
        srv.close()


def test_switch_and_threshold(orc, device):
    """layout.compact_slots: 0 never, 1 only where at least 1/32 of the rows are zero (default), 2 whenever a row is zero"""
    import chalametpir_amd as cp

    rng = np.random.default_rng(41)
    b, N, C = 9, 3 * 2048, 21
    D = random_db_matrix(rng, N, C, b)
    D[D.max(axis=1) == 0] = 1
    D[[5, 4000]] = 0  # two zero rows: below the threshold
    seed = rng.bytes(32)
    want_hint, want_dtc = orc.server_setup_from_matrix(seed, D, b)
    q = random_query(rng, N)
    want = orc.row_vector_x_compressed_transposed_matrix(q, want_dtc, N, b)[0]
    for mode, served in ((1, N), (2, N - 2), (0, N)):
        cp.tuning_set("layout.compact_slots", mode)
        srv, hint = cp.Server.setup_from_matrix(seed, D, b, device=device)
        assert srv.slots_served() == (served, N)
        assert (srv.kept_slots() is None) == (served == N)
        assert np.array_equal(hint, want_hint) and np.array_equal(srv.export_compressed(), want_dtc) and np.array_equal(srv.respond_array(q), want)
        srv.close()
    cp.tuning_set("layout.compact_slots", 1)
    D[::3] = 0
    for mode in (0, 1):
        cp.tuning_set("layout.compact_slots", mode)
        srv, _ = cp.Server.setup_from_matrix(seed, D, b, device=device)
        assert (srv.slots_served()[0] < N) == (mode == 1)
        srv.close()
    cp.tuning_set("layout.compact_slots", 1)
    # an all-zero matrix keeps every slot (nothing to gather onto)
    Z = np.zeros((600, 4), dtype=np.uint32)
    srv, hint = cp.Server.setup_from_matrix(seed, Z, b, device=device)
    assert srv.slots_served() == (600, 600) and not hint.any() and not srv.respond_array(random_query(rng, 600)).any()
    srv.close()


def test_fields_above_b_bits_do_not_keep_a_row_but_count_in_the_hint(orc, device):
    """row_wise_compress masks to b bits (matrix.rs:121) while the hint multiplies D as it is (server.rs:61): a row whose entries are
    multiples of 2^b is a zero row for respond and a non-zero row for the hint"""
    import chalametpir_amd as cp

    rng = np.random.default_rng(43)
    b, N, C = 9, 3 * 1024 + 1, 10
    D, kept = holey_matrix(rng, N, C, b, zero_frac=0.2)
    ghost = np.setdiff1d(np.arange(N), kept)[:50]
    D[ghost] = (rng.integers(1, 1 << 5, size=(ghost.size, C), dtype=np.uint64) << b).astype(np.uint32)
    seed = rng.bytes(32)
    srv, hint = cp.Server.setup_from_matrix(seed, D, b, device=device)
    want_hint, want_dtc = orc.server_setup_from_matrix(seed, D, b)
    assert srv.slots_served() == (kept.size, N)
    assert np.array_equal(hint, want_hint) and np.array_equal(srv.export_compressed(), want_dtc)
    q = random_query(rng, N)
    assert np.array_equal(srv.respond_array(q), orc.row_vector_x_compressed_transposed_matrix(q, want_dtc, N, b)[0])
    srv.close()


@pytest.mark.parametrize("b", [9, 10, 6])
def test_device_queries_single_batched_fused_and_shards(b, orc, device):
    import torch

    import chalametpir_amd as cp
    from chalametpir_amd.distributed import shard_range, shard_unit

    rng = np.random.default_rng(9100 + b)
    C = 29
    layout = cp.dtc_layout_for(100, C, b)
    N = shard_unit(layout) * 5 + 77
    D, kept = holey_matrix(rng, N, C, b, pattern="random")
    dtc = oracle_dtc(orc, D, b)
    stream = torch.cuda.current_stream()
    D_dev = dev(D)
    whole = cp.Server.from_device_matrix(D_dev, N, C, b, device=device, stream=stream)
    assert whole.slots_served() == (kept.size, N)
    qs = np.stack([random_query(rng, N) for _ in range(17)])
    wants = np.stack([orc.row_vector_x_compressed_transposed_matrix(q, dtc, N, b)[0] for q in qs])
    q_dev = dev(qs)
    r1 = torch.empty(C, dtype=torch.int32, device="cuda")
    whole.respond_device(q_dev[3], r1, stream=stream)
    torch.cuda.synchronize()
    assert np.array_equal(host(r1), wants[3])
    for fusion in (0, 1):
        cp.tuning_set("respond.batch_fusion", fusion)
        for batch in (1, 2, 5, 8, 9, 17):
            r = torch.full((batch, C), -1, dtype=torch.int32, device="cuda")
            whole.respond_batch_device(q_dev[:batch], batch, r, stream=stream)
            torch.cuda.synchronize()
            assert np.array_equal(host(r), wants[:batch]), (fusion, batch)
    cp.tuning_set("respond.batch_fusion", 1)
    # shards: each compacts its own rows; every one reads its slice of the FULL query; partial responses add up (u32 wrap-around)
    world = 3
    total = np.zeros((4, C), dtype=np.uint32)
    seen = 0
    for rank in range(world):
        lo, hi = shard_range(N, layout, rank, world)
        shard = cp.Server.from_device_matrix(D_dev[lo:hi], hi - lo, C, b, device=device, slot_offset=lo, total_slots=N, stream=stream)
        k = np.count_nonzero((kept >= lo) & (kept < hi))
        assert shard.slots_served() == (k, hi - lo)
        seen += k
        r = torch.empty((4, C), dtype=torch.int32, device="cuda")
        shard.respond_batch_device(q_dev[:4], 4, r, stream=stream)
        torch.cuda.synchronize()
        total += host(r)
        shard.close()
    assert seen == kept.size and np.array_equal(total, wants[:4])
    whole.close()


def test_wire_buffer_at_odd_addresses(orc, device):
    """Server::respond takes `&[u8]`: the wire image may start at ANY address (matrix.rs:1001-1007 copies the element bytes out).  With a
    slot map a lone caller's query is compacted on the host straight out of `query + 8` (host_gather.cpp): its scalar and tail reads must
    not assume 4-byte alignment.  Short queries (the scalar tails) and long ones (the vector body, the polled launch), every offset mod 4."""
    import ctypes

    import chalametpir_amd as cp

    rng = np.random.default_rng(1357)
    b, C = 9, 9
    for N in (1536 * 2 + 13, 1536 * 400 + 7):
        D, kept = holey_matrix(rng, N, C, b, zero_frac=0.2)
        dtc = oracle_dtc(orc, D, b)
        srv = cp.Server.from_compressed(dtc, N, b, device=device)
        assert srv.slots_served() == (kept.size, N)
        q = random_query(rng, N)
        image = wire(q)
        want = orc.server_respond(dtc, N, b, image)
        for shift in (1, 2, 3, 5, 0):
            buf = (ctypes.c_uint8 * (len(image) + 16))()
            base = ctypes.addressof(buf)
            off = (-base) % 8 + shift  # the image starts `shift` bytes past an 8-byte boundary
            ctypes.memmove(base + off, image, len(image))
            assert srv.respond_from_address(base + off, len(image)) == want, (N, shift)
        srv.close()


def test_a_database_that_does_not_fit_twice_is_served_uncompacted(orc, device):
    """with the reference and dense64 packings compaction gathers the kept rows into a temporary of (almost) D's size while D is still
    resident; where the device has no room for that the map is dropped and the whole matrix is packed -- setup must not fail for the sake of
    an optimisation.  The planar packing needs no temporary (its pack kernel reads row keep[n] of D for slot n of the image) and compacts
    even then.  The device's free memory is taken away by an allocation of this test's own (nothing is written to it) until only the
    image fits."""
    import torch

    import chalametpir_amd as cp

    rng = np.random.default_rng(2468)
    b, C = 9, 192
    N = 1536 * 300
    D, kept = holey_matrix(rng, N, C, b, zero_frac=0.2)
    dtc = oracle_dtc(orc, D, b)
    q = random_query(rng, N)
    want = orc.row_vector_x_compressed_transposed_matrix(q, dtc, N, b)[0]
    stream = torch.cuda.current_stream()
    D_dev = dev(D)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    # (the library's scratch pool keeps a block for half a second behind its last user: one of the right size, left by the test before,
    # would BE the room this test is about to take away -- let the pool give everything back first)
    time.sleep(0.8)
    image_bytes = int(cp.dtc_layout_for(N, C, b).total_words) * 4
    gather_bytes = kept.size * C * 4
    room = image_bytes + image_bytes // 5 + (64 << 20)  # what is left to the library: the image, a margin for the map and the runtime
    assert gather_bytes > room + (32 << 20)  # (so that "the image fits, the temporary does not" is a wide target at every packing)
    # (everything this test allocates through torch exists BEFORE the ballast, and the ballast is a whole number of 2 MiB pages: torch's
    # caching allocator would otherwise carve a later small tensor out of the ballast's segment and could not give the segment back)
    r = torch.empty(C, dtype=torch.int32, device="cuda")
    q_dev = dev(q)
    torch.cuda.synchronize()
    free, _ = torch.cuda.mem_get_info()
    ballast = None
    try:
        ballast = torch.empty((free - room) // (2 << 20) * (2 << 20), dtype=torch.uint8, device="cuda")
        free_now, _ = torch.cuda.mem_get_info()
        assert free_now < gather_bytes
        srv = cp.Server.from_device_matrix(D_dev, N, C, b, device=device, stream=stream)
        if srv.layout.packing == 2:
            # planar packing: the pack kernel reads the kept rows through the map, no temporary -- the database is compacted all the same
            assert srv.slots_served() == (kept.size, N)
        else:
            assert srv.slots_served() == (N, N)  # the map was dropped: every slot is resident
        srv.respond_device(q_dev, r, stream=stream)
        torch.cuda.synchronize()
        assert np.array_equal(host(r), want)
        srv.close()
    finally:
        del ballast
        torch.cuda.empty_cache()
    assert torch.cuda.mem_get_info()[0] > gather_bytes + 4 * image_bytes, "the test's own ballast was not given back"
    # with room the same matrix IS compacted
    srv = cp.Server.from_device_matrix(D_dev, N, C, b, device=device, stream=stream)
    assert srv.slots_served() == (kept.size, N)
    srv.close()


def test_host_queries_lone_and_concurrent(orc, device):
    """cpir_server_respond on host buffers with a slot map: a lone caller (pageable and page-locked: compacted on the host into the pinned
    block the kernel reads in place; long enough for the polled launch), 6 concurrent callers (uploaded whole, gathered on the device),
    the in-place path switched off, and the same answers from a server without a map"""
    import chalametpir_amd as cp

    rng = np.random.default_rng(77)
    b, C = 9, 12
    N = 1536 * 500 + 11  # > 2^19 kept words: the lone pageable caller takes the polled launch where the packing offers it
    D, kept = holey_matrix(rng, N, C, b, zero_frac=0.11)
    dtc = oracle_dtc(orc, D, b)
    srv = cp.Server.setup_from_matrix(rng.bytes(32), D, b, device=device)[0]
    assert srv.slots_served() == (kept.size, N) and kept.size > (1 << 19)
    cp.tuning_set("layout.compact_slots", 0)
    plain = cp.Server.from_compressed(dtc, N, b, device=device)
    cp.tuning_set("layout.compact_slots", 1)
    assert plain.slots_served() == (N, N)
    qs = [random_query(rng, N) for _ in range(6)]
    wants = [orc.row_vector_x_compressed_transposed_matrix(q, dtc, N, b)[0] for q in qs]
    pin = cp.PinnedArray(N)
    try:
        for rnd in range(3):
            for q, want in zip(qs, wants):
                assert np.array_equal(srv.respond_array(q), want)  # lone, pageable
                pin.array[:] = q
                assert np.array_equal(srv.respond_array(pin.array), want)  # lone, page-locked
                assert np.array_equal(plain.respond_array(q), want)
            cp.tuning_set("respond.host_zero_copy", rnd % 2)  # second round: upload first, gather on the device
            cp.tuning_set("respond.host_fill_timeout_us", 0 if rnd == 2 else 2000)  # third round: two launches instead of polling
        cp.tuning_set("respond.host_zero_copy", 1)
        cp.tuning_set("respond.host_fill_timeout_us", 2000)
        errors = []

        def caller(k):
            try:
                for i in range(12):
                    j = (k + i) % len(qs)
                    if not np.array_equal(srv.respond_array(qs[j]), wants[j]):
                        errors.append((k, i))
            except Exception as exc:  # noqa: BLE001
                errors.append(repr(exc))

        ts = [threading.Thread(target=caller, args=(k,)) for k in range(6)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert not errors, errors[:3]
        # two to four callers at a time: in-place rounds (respond.inplace_seats) -- with a map a pageable caller compacts its query into its
        # seat of the pinned block while the pass polls the seats' progress; a page-locked one (caller 1 here) stays on the upload path
        pins = [cp.PinnedArray(N) for _ in range(2)]
        for pa, q in zip(pins, qs):
            pa.array[:] = q
        before = srv.host_path_counts()
        for crew in (2, 3, 4):
            start = threading.Barrier(crew)

            def few(k):
                start.wait()
                try:
                    for i in range(10):
                        j = (k + i) % len(qs)
                        got = srv.respond_array(pins[j].array if (k == 1 and j < 2) else qs[j])
                        if not np.array_equal(got, wants[j]):
                            errors.append((crew, k, i))
                except Exception as exc:  # noqa: BLE001
                    errors.append(repr(exc))

            ts = [threading.Thread(target=few, args=(k,)) for k in range(crew)]
            [t.start() for t in ts]
            [t.join() for t in ts]
            assert not errors, errors[:3]
        after = srv.host_path_counts()
        assert after["calls"] - before["calls"] == 90 and after["polled_passes_given_up"] == before["polled_passes_given_up"], (before, after)
        if srv.layout.packing == 2:  # planar
            assert after["in_place_rounds"] > before["in_place_rounds"], (before, after)
        else:
            assert after["in_place_rounds"] == 0, after
        for pa in pins:
            pa.close()
        clone = srv.clone()
        assert np.array_equal(clone.respond_array(qs[0]), wants[0])
        clone.close()
    finally:
        pin.close()
        srv.close()
        plain.close()


def test_group_handle_with_compacted_shards(orc, device, group_devices):
    """cpir_server_setup_multi: every shard finds its own zero rows; host queries are scattered, partial responses summed on the host"""
    import chalametpir_amd as cp

    rng = np.random.default_rng(78)
    b, C = 9, 15
    N = 21504 * 3 + 100  # three shard units of every packing (dense64 at b = 9: lcm(7 * 1024, 3) slots)
    D, kept = holey_matrix(rng, N, C, b, pattern="runs")
    seed = rng.bytes(32)
    want_hint, want_dtc = orc.server_setup_from_matrix(seed, D, b)
    grp, hint = cp.Server.setup_from_matrix(seed, D, b, devices=group_devices(3))
    try:
        shards = grp.group_shards()
        assert len(shards) == 3
        # the 1/32 rule is applied shard by shard: the last shard's 300 zero rows of 21 604 stay in its image
        expect = 0
        for _, lo, cnt in shards:
            k = int(np.count_nonzero((kept >= lo) & (kept < lo + cnt)))
            expect += k if (cnt - k) * 32 >= cnt else cnt
        assert expect < N and grp.slots_served() == (expect, N)
        assert np.array_equal(hint, want_hint) and np.array_equal(grp.export_compressed(), want_dtc)
        for _ in range(4):
            q = random_query(rng, N)
            assert np.array_equal(grp.respond_array(q), orc.row_vector_x_compressed_transposed_matrix(q, want_dtc, N, b)[0])
    finally:
        grp.close()


@pytest.mark.parametrize("arity", [3, 4])
def test_key_value_setup_serves_exactly_the_owned_slots(arity, orc, device):
    """Server::setup on a key-value database: the kept slots are the n slots the filter assigns to the n keys (matrix.rs:727-740), the
    packed database, hint, filter bytes and responses equal the oracle's"""
    import chalametpir_amd as cp

    rng = np.random.default_rng(500 + arity)
    n = 3000
    db = {}
    while len(db) < n:
        db[rng.bytes(24)] = rng.bytes(int(rng.integers(1, 80)))
    seed, fseeds = rng.bytes(32), rng.bytes(3200)
    srv, hint_bytes, filter_bytes = cp.Server.setup(seed, db, arity, device=device, filter_seed_material=fseeds)
    keys, vals = list(db.keys()), list(db.values())
    b = orc.find_encoded_db_matrix_element_bit_length(n)
    D, filt, _ = orc.from_kv_database(arity, keys, vals, b, fseeds)
    N = D.shape[0]
    owned = np.nonzero(D.max(axis=1) != 0)[0]
    assert owned.size == n  # every key's row carries the digest of its key: never all zero
    assert srv.slots_served() == (n, N) and np.array_equal(srv.kept_slots(), owned)
    want_hint, want_dtc = orc.server_setup_from_matrix(seed, D, b)
    assert filter_bytes == filt.to_bytes() and hint_bytes == wire(want_hint)
    assert np.array_equal(srv.export_compressed(), want_dtc)
    q = random_query(rng, N)
    assert np.array_equal(srv.respond_array(q), orc.row_vector_x_compressed_transposed_matrix(q, want_dtc, N, b)[0])
    srv.close()


@pytest.mark.parametrize("b,pattern", [(9, "random"), (10, "runs"), (6, "alternate"), (13, "random")])
def test_import_of_a_compressed_database_leaves_the_empty_rows_out_too(b, pattern, orc, device):
    """cpir_server_from_compressed (the database arrives in the reference's compressed form, `matrix.rs:98-205`): which slots hold something
    is read off the OR of the compressed rows, the kept slots' fields are gathered into a compressed matrix of their own and that is
    imported -- same map, same image, same answers as the server that was set up from the matrix; dirty bits beyond the b significant bits
    of a field (and beyond the cf fields of a word) neither keep a row nor survive the export"""
    import torch

    import chalametpir_amd as cp

    rng = np.random.default_rng(7100 + b)
    cf = cf_of(b)
    stream = torch.cuda.current_stream()
    for N, C in ((cf * 1024 * 2 + 7, 29), (3000, 5)):
        D, kept = holey_matrix(rng, N, C, b, pattern=pattern)
        want_dtc = oracle_dtc(orc, D, b)
        dirty = want_dtc.copy()
        S = 32 // cf
        if S > b:  # bits above the field inside its slot
            dirty |= np.uint32(1 << (S - 1))
        if cf * S < 32:  # bits above the last slot of a word
            dirty |= np.uint32(1 << 31)
        for src in (want_dtc, dirty):
            srv = cp.Server.from_compressed(src, N, b, device=device)
            assert srv.slots_served() == (kept.size, N) and np.array_equal(srv.kept_slots(), kept)
            assert np.array_equal(srv.export_compressed(), want_dtc)
            Q = np.stack([random_query(rng, N) for _ in range(14)])
            want = np.stack([orc.row_vector_x_compressed_transposed_matrix(q, want_dtc, N, b)[0] for q in Q])
            assert np.array_equal(srv.respond_array(Q[0]), want[0])
            R = torch.full((14, C), -1, dtype=torch.int32, device="cuda")
            srv.respond_batch_device(dev(Q), 14, R, stream=stream)
            torch.cuda.synchronize()
            assert np.array_equal(host(R), want)
            srv.close()
    cp.tuning_set("layout.compact_slots", 0)
    srv = cp.Server.from_compressed(want_dtc, N, b, device=device)
    assert srv.slots_served() == (N, N) and np.array_equal(srv.respond_array(Q[1]), want[1])
    srv.close()
    cp.tuning_set("layout.compact_slots", 1)
