"""GPU parity of GROUP handles: one process, the database split along the filter slots over several devices
(cpir_server_setup_multi / cpir_server_setup_kv_multi), host queries scattered and partial responses summed on the host.

The reference has no multi-device code; the property is that a group is indistinguishable from one device: same hint, same
packed database, same responses, bit for bit.  The device list of every group comes from the `group_devices` fixture: DISTINCT devices
wherever the box has them (shard i on device i mod the visible devices -- on the driver's 8-GPU node every test below runs across
devices: peer copies, per-device streams, the root's sum behind other devices' events), the one GPU listed several times on a one-GPU box
(every shard still has its own packed image, streams, query slice and partial response).  Nothing here skips."""
import threading

import numpy as np
import pytest

from _cases import cf_of, random_db_matrix, random_query, wire

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("b,N,C,shards", [(9, 3 * 1536 + 700, 37, 3), (10, 9 * 1536, 64, 2), (13, 5 * 512 + 1, 21, 4), (6, 4 * 4096 + 5, 33, 3),
                                          (9, 1000, 5, 8)])
def test_group_setup_and_respond_match_the_oracle(b, N, C, shards, orc, device, group_devices):
    import chalametpir_amd as cp

    rng = np.random.default_rng(b * 100 + shards)
    seed = rng.bytes(32)
    D = random_db_matrix(rng, N, C, b)
    want_hint, want_dtc = orc.server_setup_from_matrix(seed, D, b)
    srv, hint = cp.Server.setup_from_matrix(seed, D, b, devices=group_devices(shards))
    parts = srv.group_shards()
    assert 1 <= len(parts) <= shards and parts[0][1] == 0 and sum(p[2] for p in parts) == N
    cf = cf_of(b)
    for (_, lo, cnt), nxt in zip(parts, parts[1:] + [(0, N, 0)]):
        assert cnt > 0 and lo + cnt == nxt[1] and lo % cf == 0  # contiguous, non-empty, word-aligned
    if N >= 8 * 1536:
        assert len(parts) == shards
    assert np.array_equal(hint, want_hint)  # sum of the per-shard partial hints
    assert np.array_equal(srv.export_compressed(), want_dtc)  # stitched from the shards
    assert (srv.decompressed_num_cols, srv.mat_elem_bit_len, srv.response_len) == (N, b, C)
    for _ in range(3):
        q = random_query(rng, N)
        assert srv.respond(wire(q)) == orc.server_respond(want_dtc, N, b, wire(q))
    # wire errors are those of one device (matrix.rs:973-1010, 329-331)
    with pytest.raises(cp.ChalametPIRError) as e:
        srv.respond(wire(random_query(rng, N + 1)))
    assert e.value.variant == "IncompatibleDimensionForRowVectorTransposedMatrixMultiplication"
    # caller-supplied A, and a clone sharing the shards
    A = orc.generate_from_seed(1774, N, seed)
    srv2, hint2 = cp.Server.setup_from_matrix(seed, D, b, pub_mat_a=A, devices=group_devices(shards))
    assert np.array_equal(hint2, want_hint)
    twin = srv.clone()
    srv.close()
    q = random_query(rng, N)
    assert twin.respond(wire(q)) == srv2.respond(wire(q)) == orc.server_respond(want_dtc, N, b, wire(q))


def test_group_setup_with_empty_rows_gives_the_same_hint_every_time(orc, device, group_devices):
    """cpir_server_setup_multi on a matrix with empty rows (every shard compacts; its hint product then takes the byte-plane split of its
    rows of D, one product per shard back to back): forty setups, every hint equal to the single-device hint.  (One in a dozen was wrong in
    every entry while the products' scratch came from the runtime's stream-ordered allocator: tests/test_gpu_setup.py, the test above the
    matrix-core one.)"""
    import chalametpir_amd as cp

    rng = np.random.default_rng(9728)
    N, C, b = 3 * 1536 * 2 + 512, 20, 9
    D = random_db_matrix(rng, N, C, b)
    D[rng.random(N) < 0.2] = 0
    seed = rng.bytes(32)
    one, hint1 = cp.Server.setup_from_matrix(seed, D, b, device=device)
    one.close()
    for it in range(40):
        grp, hint = cp.Server.setup_from_matrix(seed, D, b, devices=group_devices(3))
        assert np.array_equal(hint, hint1), it
        grp.close()


def test_group_from_kv_database_equals_one_device(orc, device, group_devices):
    import chalametpir_amd as cp

    rng = np.random.default_rng(4242)
    n = 3000
    db = {}
    while len(db) < n:
        db[rng.bytes(int(rng.integers(16, 33)))] = rng.bytes(int(rng.integers(1, 200)))
    seed, fseeds = rng.bytes(32), rng.bytes(32 * 100)
    for arity in (3, 4):
        one, hint1, filt1 = cp.Server.setup(seed, db, arity, device=device, filter_seed_material=fseeds)
        grp, hint2, filt2 = cp.Server.setup(seed, db, arity, devices=group_devices(3), filter_seed_material=fseeds)
        assert hint1 == hint2 and filt1 == filt2 and len(grp.group_shards()) >= 1
        assert np.array_equal(one.export_compressed(), grp.export_compressed())
        N = one.decompressed_num_cols
        for _ in range(4):
            q = wire(random_query(rng, N))
            assert one.respond(q) == grp.respond(q)
        ph = grp.setup_timings()
        assert ph["total"] > 0 and ph["encode"] > 0


def test_group_respond_is_reentrant(orc, device, group_devices):
    import chalametpir_amd as cp

    rng = np.random.default_rng(99)
    N, C, b = 6 * 1536 + 11, 29, 9
    D = random_db_matrix(rng, N, C, b)
    srv, _ = cp.Server.setup_from_matrix(rng.bytes(32), D, b, devices=group_devices(4))
    dtc = orc.row_wise_compress(orc.transpose(D), b)
    qs = [random_query(rng, N) for _ in range(16)]
    wants = [orc.row_vector_x_compressed_transposed_matrix(q, dtc, N, b)[0] for q in qs]
    errors = []

    def work(k):
        try:
            for i in range(12):
                j = (k + i) % len(qs)
                if not np.array_equal(srv.respond_array(qs[j]), wants[j]):
                    errors.append((k, j))
        except Exception as exc:  # noqa: BLE001
            errors.append(repr(exc))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(12)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors
    # every shard is an ordinary server asked through its own respond: each answered -- and counted -- every one of the 144 queries
    counts = srv.host_path_counts()
    assert counts["calls"] == 4 * 144, counts
    assert counts["calls"] == counts["alone"] + counts["in_uploaded_rounds"] + counts["in_in_place_rounds"], counts


def test_device_pointer_entry_points_take_a_group_on_the_default_stream(orc, device, group_devices):
    """(round 3 rejected a group here; since round 4 the exchange runs on the devices: see the peer-exchange test below) -- the NULL
    stream and a tiny database"""
    import torch

    import chalametpir_amd as cp

    rng = np.random.default_rng(5)
    N, C, b = 4 * 1536, 8, 9
    D = random_db_matrix(rng, N, C, b)
    srv, _ = cp.Server.setup_from_matrix(rng.bytes(32), D, b, devices=group_devices(2))
    dtc = orc.row_wise_compress(orc.transpose(D), b)
    qh = random_query(rng, N)
    q = torch.from_numpy(qh.view(np.int32)).cuda()
    r = torch.zeros(C, dtype=torch.int32, device="cuda")
    srv.respond_device(q, r)
    torch.cuda.synchronize()
    want = orc.row_vector_x_compressed_transposed_matrix(qh, dtc, N, b)[0]
    assert np.array_equal(r.cpu().numpy().view(np.uint32), want)
    r.zero_()
    srv.respond_batch_device(q, 1, r)
    torch.cuda.synchronize()
    assert np.array_equal(r.cpu().numpy().view(np.uint32), want)
    srv.close()


def test_lifecycle_stress_servers_groups_and_setups_from_many_threads(orc, device, group_devices):
    """handles created, cloned, queried and dropped concurrently from several threads -- single-device servers (coalescing arenas),
    group handles (per-shard worker threads) and full setups (background release of A / D): every response must still be exact and
    nothing may hang or crash"""
    import chalametpir_amd as cp

    rng = np.random.default_rng(31337)
    N, C, b = 4 * 1536 + 9, 23, 9
    D = random_db_matrix(rng, N, C, b)
    dtc = orc.row_wise_compress(orc.transpose(D), b)
    seed = rng.bytes(32)
    want_hint, _ = orc.server_setup_from_matrix(seed, D, b)
    qs = [random_query(rng, N) for _ in range(8)]
    wants = [orc.row_vector_x_compressed_transposed_matrix(q, dtc, N, b)[0] for q in qs]
    errors = []

    def worker(k):
        try:
            for it in range(6):
                kind = (k + it) % 3
                if kind == 0:
                    srv = cp.Server.from_compressed(dtc, N, b, device=device)
                elif kind == 1:
                    srv, hint = cp.Server.setup_from_matrix(seed, D, b, devices=group_devices(2 + it % 3))
                    if not np.array_equal(hint, want_hint):
                        errors.append(("group hint", k, it))
                else:
                    srv, hint = cp.Server.setup_from_matrix(seed, D, b, device=device)
                    if not np.array_equal(hint, want_hint):
                        errors.append(("hint", k, it))
                twin = srv.clone()
                for j in range(4):
                    i = (k + it + j) % len(qs)
                    h = twin if j % 2 else srv
                    if not np.array_equal(h.respond_array(qs[i]), wants[i]):
                        errors.append(("respond", k, it, j))
                srv.close()
                if not np.array_equal(twin.respond_array(qs[0]), wants[0]):  # the clone keeps the database alive
                    errors.append(("clone", k, it))
                twin.close()
        except Exception as exc:  # noqa: BLE001
            errors.append(repr(exc))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(6)]
    [t.start() for t in threads]
    [t.join(600) for t in threads]
    assert not any(t.is_alive() for t in threads), "a worker hung"
    assert not errors, errors[:5]


@pytest.mark.parametrize("b,holes", [(9, False), (9, True), (12, True)])
def test_group_answers_device_queries_with_a_peer_exchange(b, holes, orc, device, group_devices):
    """cpir_server_respond_device / _batch_device on a group handle: q and r on the root device, every shard pulls its slots over the
    peer link, the C-word partials are pushed into the root's table and summed by a kernel on the caller's stream -- same responses as
    the oracle on the whole matrix, for one query, for batches around and beyond the 48-query round, on databases with and without rows
    that are left out of the image (compact.hip)"""
    import torch

    import chalametpir_amd as cp

    rng = np.random.default_rng(6000 + b + holes)
    C, N = 23, 21504 * 3 + 901
    D = random_db_matrix(rng, N, C, b)
    if holes:
        D[rng.random(N) < 0.15] = 0
    seed = rng.bytes(32)
    _, want_dtc = orc.server_setup_from_matrix(seed, D, b)
    devs = group_devices(3)
    assert len({d.ordinal for d in devs}) == min(3, group_devices.visible)  # (distinct devices wherever the box has them)
    grp, _ = cp.Server.setup_from_matrix(seed, D, b, devices=devs)
    try:
        assert len(grp.group_shards()) == 3 and [p[0] for p in grp.group_shards()] == [d.ordinal for d in devs]
        if holes:
            assert grp.slots_served()[0] < N
        stream = torch.cuda.current_stream()
        qs = np.stack([random_query(rng, N) for _ in range(70)])
        wants = np.stack([orc.row_vector_x_compressed_transposed_matrix(q, want_dtc, N, b)[0] for q in qs[:60]])
        q_dev = torch.from_numpy(qs.view(np.int32)).cuda()
        r1 = torch.full((C,), -1, dtype=torch.int32, device="cuda")
        grp.respond_device(q_dev[7], r1, stream=stream)
        torch.cuda.synchronize()
        assert np.array_equal(r1.cpu().numpy().view(np.uint32), wants[7])
        for batch in (1, 2, 9, 32, 33, 40, 48, 49, 60):
            r = torch.full((batch, C), -1, dtype=torch.int32, device="cuda")
            grp.respond_batch_device(q_dev[:batch], batch, r, stream=stream)
            torch.cuda.synchronize()
            assert np.array_equal(r.cpu().numpy().view(np.uint32), wants[:batch]), batch
        # calls enqueued back to back on two streams (every call takes the next of four contexts; a context is reused behind its own sum)
        s2 = torch.cuda.Stream()
        outs = []
        for i in range(10):
            r = torch.empty((4, C), dtype=torch.int32, device="cuda")
            st = stream if i % 2 == 0 else s2
            with torch.cuda.stream(st):
                grp.respond_batch_device(q_dev[4 * (i % 8):4 * (i % 8) + 4], 4, r, stream=st)
            outs.append((i, r))
        torch.cuda.synchronize()
        for i, r in outs:
            assert np.array_equal(r.cpu().numpy().view(np.uint32), wants[4 * (i % 8):4 * (i % 8) + 4]), i
        # and the host entry point of the same handle still agrees
        assert np.array_equal(grp.respond_array(qs[3]), wants[3])
    finally:
        grp.close()
