"""GPU parity of the WIDE pass (respond_planar_wide_kernel): up to 24 queries answered by one stream of the packed database, one 8-wave
block per CU, the A row sets walked in a loop, single-buffered fragments.

The reference answers one query at a time (`server.rs:184-190` -> `matrix.rs:328-485`); the property is that every query of a fused
batch gets exactly the response the oracle computes for it alone, whatever the batch size, the width of the database, the alignment of
the query buffers or the number of passes in the launch."""
import numpy as np
import pytest

from _cases import cf_of, random_db_matrix, random_query

pytestmark = pytest.mark.gpu


def _responses(orc, Q, dtc, N, b):
    return np.stack([orc.row_vector_x_compressed_transposed_matrix(q, dtc, N, b)[0] for q in Q])


@pytest.mark.parametrize("b", [4, 8, 9, 10, 11, 12, 13, 14])
def test_every_batch_size_through_the_wide_pass(b, orc, device):
    """every fused batch runs on the wide kernel (1 .. 24 queries per pass: one to six row sets, the last one partly filled; larger
    batches in as few passes as 24 each allow), and so does every unfused one (passes of ONE query, in slice and in interleaved order);
    respond.ks_major = 2 answers the same batches in passes of 4 on the step-major kernel; every plane count (b = 4 .. 14); N ragged (the
    last step is guarded), more than one visit per block"""
    import torch

    import chalametpir_amd as cp

    rng = np.random.default_rng(2400 + b)
    stream = torch.cuda.current_stream()
    N, C = 5 * 512 * 3 + 129, 150  # 16 steps, the last with 129 valid slots; 10 column tiles = 2 groups of 8
    D = random_db_matrix(rng, N, C, b)
    dtc = orc.row_wise_compress(orc.transpose(D), b)
    srv = cp.Server.from_compressed(dtc, N, b, device=device)
    nq = 50
    Q = np.stack([random_query(rng, N) for _ in range(nq)])
    want = _responses(orc, Q, dtc, N, b)
    Q_dev = torch.from_numpy(Q.view(np.int32)).cuda()
    try:
        for ks_major, fusion, order in ((1, 1, -1), (1, 0, 0), (1, 0, 1), (1, 1, 1), (2, 1, -1), (2, 0, 0)):
            cp.tuning_set("respond.ks_major", ks_major)
            cp.tuning_set("respond.batch_fusion", fusion)
            cp.tuning_set("respond.interleave_passes", order)
            for k in (list(range(1, 27)) + [31, 32, 33, 47, 48, 49, 50] if (ks_major, fusion) == (1, 1) else [1, 2, 3, 4, 5, 7, 8, 9, 12, 13, 24, 25, 37, 50]):
                R = torch.full((k, C), -1, dtype=torch.int32, device="cuda")
                srv.respond_batch_device(Q_dev, k, R, stream=stream)
                torch.cuda.synchronize()
                assert np.array_equal(R.cpu().numpy().view(np.uint32), want[:k]), (b, ks_major, fusion, order, k)
    finally:
        cp.tuning_reset()
        srv.close()


@pytest.mark.parametrize("b,steps,C,passes", [(9, 1, 17, 7), (9, 9, 130, 3), (10, 40, 300, 70), (8, 700, 64, 33), (12, 23, 129, 257), (9, 300, 940, 32)])
def test_passes_in_interleaved_order_share_out_the_pass_unit_space(b, steps, C, passes, orc, device):
    """the interleaved order of a launch of many passes (what every multi-GPU shard runs): the (pass, unit) space is split evenly over the
    blocks, so a block may hold the tail of one pass, whole passes and the head of another -- fewer units than blocks, more passes than
    blocks per XCD, one step only, ragged last steps; every query's response equals the oracle's, and the slice order's"""
    import torch

    import chalametpir_amd as cp

    rng = np.random.default_rng(7000 + steps)
    stream = torch.cuda.current_stream()
    N = steps * 512 - int(rng.integers(0, 500))
    D = random_db_matrix(rng, N, C, b)
    dtc = orc.row_wise_compress(orc.transpose(D), b)
    srv = cp.Server.from_compressed(dtc, N, b, device=device)
    distinct = min(passes, 12)  # (the oracle answers a dozen distinct queries; the launch repeats them)
    Q = np.stack([random_query(rng, N) for _ in range(distinct)])
    want = _responses(orc, Q, dtc, N, b)
    pick = np.arange(passes) % distinct
    Q_dev = torch.from_numpy(Q[pick].view(np.int32)).cuda()
    try:
        cp.tuning_set("respond.batch_fusion", 0)
        got = {}
        for order in (1, 0):
            cp.tuning_set("respond.interleave_passes", order)
            R = torch.full((passes, C), -1, dtype=torch.int32, device="cuda")
            srv.respond_batch_device(Q_dev, passes, R, stream=stream)
            torch.cuda.synchronize()
            got[order] = R.cpu().numpy().view(np.uint32)
            assert np.array_equal(got[order], want[pick]), (b, steps, C, passes, order)
        # fused passes of several queries each, interleaved: 3 passes of W queries
        cp.tuning_set("respond.batch_fusion", 1)
        cp.tuning_set("respond.interleave_passes", 1)
        k = min(passes, 60)
        R = torch.full((k, C), -1, dtype=torch.int32, device="cuda")
        srv.respond_batch_device(Q_dev, k, R, stream=stream)
        torch.cuda.synchronize()
        assert np.array_equal(R.cpu().numpy().view(np.uint32), want[pick[:k]]), (b, steps, C, passes, "fused")
    finally:
        cp.tuning_reset()
        srv.close()


def test_wide_pass_guarded_queries_shards_and_extreme_words(orc, device):
    """the guarded paths: query rows that are only 4-byte aligned (N % 4 == 3: every row but the first), a buffer shifted by one word, a
    shard that starts at an odd slot and ends inside a packing unit; all-ones and all-0x80 query words and maximal database entries
    (the signed-byte split's corner values)"""
    import torch

    import chalametpir_amd as cp

    rng = np.random.default_rng(2424)
    stream = torch.cuda.current_stream()
    for b in (9, 12, 6):
        cf = cf_of(b)
        N, C = 3 * 1024 * cf + 7, 37
        D = random_db_matrix(rng, N, C, b)
        D[5] = (1 << b) - 1
        D[:, 3] = (1 << b) - 1
        dtc = orc.row_wise_compress(orc.transpose(D), b)
        srv = cp.Server.from_compressed(dtc, N, b, device=device)
        nq = 24
        Q = np.stack([random_query(rng, N) for _ in range(nq)])
        Q[1] = 0xFFFFFFFF
        Q[2] = 0x80808080
        Q[3] = 0x7F7F7F7F
        Q[4] = 0
        want = _responses(orc, Q, dtc, N, b)
        Q_dev = torch.from_numpy(Q.view(np.int32)).cuda()
        for k in (13, 17, 24):
            R = torch.full((k, C), -1, dtype=torch.int32, device="cuda")
            srv.respond_batch_device(Q_dev, k, R, stream=stream)
            torch.cuda.synchronize()
            assert np.array_equal(R.cpu().numpy().view(np.uint32), want[:k]), (b, k)
        # the whole block of queries shifted by 1..3 words
        buf = torch.zeros(nq * N + 8, dtype=torch.int32, device="cuda")
        for shift in (1, 2, 3):
            buf[shift:shift + nq * N] = Q_dev.reshape(-1)
            R = torch.full((nq, C), -1, dtype=torch.int32, device="cuda")
            srv.respond_batch_device(buf[shift:shift + nq * N].view(nq, N), nq, R, stream=stream)
            torch.cuda.synchronize()
            assert np.array_equal(R.cpu().numpy().view(np.uint32), want), (b, shift)
        lo, hi = 1027, N - 5
        D_dev = torch.from_numpy(np.ascontiguousarray(D[lo:hi]).view(np.int32)).cuda()
        shard = cp.Server.from_device_matrix(D_dev, hi - lo, C, b, device=device, slot_offset=lo, total_slots=N)
        dtc_shard = orc.row_wise_compress(orc.transpose(D[lo:hi]), b)
        want_part = _responses(orc, Q[:, lo:hi], dtc_shard, hi - lo, b)
        part = torch.full((nq, C), -1, dtype=torch.int32, device="cuda")
        shard.respond_batch_device(Q_dev, nq, part, stream=stream)
        torch.cuda.synchronize()
        assert np.array_equal(part.cpu().numpy().view(np.uint32), want_part), b
        shard.close()
        srv.close()


def test_wide_pass_over_column_windows(orc, device):
    """24 queries x 1 024 columns of accumulators fill the CU's LDS: a wider database is answered window by window (whole groups of 8
    column tiles), each a launch over all steps; 3 100 columns = 194 tiles = 25 groups -> 4 windows at 24 queries, fewer for fewer"""
    import torch

    import chalametpir_amd as cp

    rng = np.random.default_rng(3100)
    stream = torch.cuda.current_stream()
    for b, N, C, holes in ((9, 2 * 512 + 77, 3100, False), (6, 700, 1601, False), (9, 3 * 512 + 5, 2100, True)):
        D = random_db_matrix(rng, N, C, b)
        if holes:  # ... and with rows left out of the image: the slot map applied in the kernel, window by window
            D[rng.random(N) < 0.25] = 0
        dtc = orc.row_wise_compress(orc.transpose(D), b)
        srv = cp.Server.from_compressed(dtc, N, b, device=device)
        assert (srv.slots_served()[0] < N) == holes
        nq = 48
        Q = np.stack([random_query(rng, N) for _ in range(nq)])
        want = _responses(orc, Q, dtc, N, b)
        Q_dev = torch.from_numpy(Q.view(np.int32)).cuda()
        for k in (13, 14, 16, 20, 24, 25, 36, 48):
            R = torch.full((k, C), -1, dtype=torch.int32, device="cuda")
            srv.respond_batch_device(Q_dev, k, R, stream=stream)
            torch.cuda.synchronize()
            assert np.array_equal(R.cpu().numpy().view(np.uint32), want[:k]), (b, k)
        srv.close()


def test_wide_pass_on_a_compacted_database_and_a_group(orc, device, group_devices):
    """rows that hold nothing are left out of the image (compact.hip): the queries are gathered onto the kept slots in front of the wide
    pass; a group of shards answers the same batch through its device exchange"""
    import torch

    import chalametpir_amd as cp

    rng = np.random.default_rng(777)
    stream = torch.cuda.current_stream()
    b, C, N = 9, 61, 21504 * 2 + 333
    D = random_db_matrix(rng, N, C, b)
    D[rng.random(N) < 0.2] = 0
    seed = rng.bytes(32)
    _, dtc = orc.server_setup_from_matrix(seed, D, b)
    srv, _ = cp.Server.setup_from_matrix(seed, D, b, device=device)
    grp, _ = cp.Server.setup_from_matrix(seed, D, b, devices=group_devices(2))
    try:
        assert srv.slots_served()[0] < N
        nq = 30
        Q = np.stack([random_query(rng, N) for _ in range(nq)])
        want = _responses(orc, Q, dtc, N, b)
        Q_dev = torch.from_numpy(Q.view(np.int32)).cuda()
        for h in (srv, grp):
            for k in (13, 24, 30):
                R = torch.full((k, C), -1, dtype=torch.int32, device="cuda")
                h.respond_batch_device(Q_dev, k, R, stream=stream)
                torch.cuda.synchronize()
                assert np.array_equal(R.cpu().numpy().view(np.uint32), want[:k]), k
    finally:
        srv.close()
        grp.close()


def test_wide_pass_large_batches_and_their_remainders(orc, device):
    """batches far beyond one launch's passes: as few passes as 24 queries each allow, all of the same width, and what is left over as
    a launch of its own -- 241 = 10 x 22 + 21, 1 009 = 42 x 24 + 1 (a remainder of ONE query), 100 = 5 x 20; the responses are those of
    100 distinct queries repeated"""
    import torch

    import chalametpir_amd as cp

    rng = np.random.default_rng(1009)
    stream = torch.cuda.current_stream()
    b, N, C = 9, 2048 + 300, 70
    D = random_db_matrix(rng, N, C, b)
    dtc = orc.row_wise_compress(orc.transpose(D), b)
    srv = cp.Server.from_compressed(dtc, N, b, device=device)
    base = np.stack([random_query(rng, N) for _ in range(100)])
    want = _responses(orc, base, dtc, N, b)
    layout = srv.physical_layout
    assert [cp.respond_batch_pass_width(layout, k) for k in (1, 4, 5, 24, 25, 32, 48, 100, 241, 1009)] == [1, 4, 5, 24, 13, 16, 24, 20, 22, 24]
    for k in (100, 241, 1009):
        idx = np.arange(k) % 100
        Q_dev = torch.from_numpy(base[idx].view(np.int32)).cuda()
        R = torch.full((k, C), -1, dtype=torch.int32, device="cuda")
        srv.respond_batch_device(Q_dev, k, R, stream=stream)
        torch.cuda.synchronize()
        assert np.array_equal(R.cpu().numpy().view(np.uint32), want[idx]), k
    srv.close()


def test_wide_pass_random_shapes(orc, device):
    """seeded random shapes through the wide pass: bit length, slots around the block count of the one-block-per-CU grid (fewer steps
    than blocks, ragged last steps), columns around the 8-tile groups, shard windows that start at odd slots, query blocks shifted off
    their 16-byte alignment, any batch size up to 30, with and without rows left out of the image -- against the oracle"""
    import torch

    import chalametpir_amd as cp

    rng = np.random.default_rng(2424_1)
    stream = torch.cuda.current_stream()
    for trial in range(36):
        b = int(rng.choice([4, 6, 8, 9, 9, 10, 12, 14]))
        cf = cf_of(b)
        steps = int(rng.choice([1, 2, 3, 7, 31, 200, 257, 300]))
        N = max(cf, steps * 512 - int(rng.integers(0, 512)))
        C = int(rng.choice([1, 3, 16, 17, 127, 128, 129, 260]))
        if N * C > 6_000_000:
            C = max(1, 6_000_000 // N)
        D = random_db_matrix(rng, N, C, b)
        holes = trial % 3 == 0
        if holes:
            D[rng.random(N) < 0.2] = 0
        lo = int(rng.integers(0, max(1, N // 3)))
        hi = N - int(rng.integers(0, max(1, N // 5)))
        if rng.integers(0, 3) == 0:
            lo, hi = 0, N
        dtc = orc.row_wise_compress(orc.transpose(D[lo:hi]), b)
        D_dev = torch.from_numpy(np.ascontiguousarray(D[lo:hi]).view(np.int32)).cuda()
        srv = cp.Server.from_device_matrix(D_dev, hi - lo, C, b, device=device, slot_offset=lo, total_slots=N)
        k = int(rng.integers(1, 31))
        Q = np.stack([random_query(rng, N) for _ in range(k)])
        want = np.stack([orc.row_vector_x_compressed_transposed_matrix(q[lo:hi], dtc, hi - lo, b)[0] for q in Q])
        shift = int(rng.integers(0, 4))
        buf = torch.zeros(k * N + 8, dtype=torch.int32, device="cuda")
        buf[shift:shift + k * N] = torch.from_numpy(Q.view(np.int32)).cuda().reshape(-1)
        R = torch.full((k, C), -1, dtype=torch.int32, device="cuda")
        srv.respond_batch_device(buf[shift:shift + k * N].view(k, N), k, R, stream=stream)
        torch.cuda.synchronize()
        assert np.array_equal(R.cpu().numpy().view(np.uint32), want), (trial, b, N, C, lo, hi, shift, k, holes)
        r = torch.full((C,), -1, dtype=torch.int32, device="cuda")
        srv.respond_device(buf[shift:shift + N], r, stream=stream)  # one query, one pass: the wide kernel with one row set
        torch.cuda.synchronize()
        assert np.array_equal(r.cpu().numpy().view(np.uint32), want[0]), (trial, "lone")
        srv.close()


def test_random_shapes_passes_and_orders(orc, device):
    """seeded random shapes through the wide kernel's pass bookkeeping: steps from 1 to a few hundred (fewer units than blocks, the XCD split
    on and off), column counts around the 8-tile groups, 1 .. 70 passes of 1 .. 6 queries, both orders, shards with an offset -- every
    response against the oracle"""
    import torch

    import chalametpir_amd as cp

    rng = np.random.default_rng(20251004)
    stream = torch.cuda.current_stream()
    try:
        for trial in range(36):
            b = int(rng.choice([9, 9, 10, 8, 12]))
            steps = int(rng.choice([1, 2, 5, 7, 8, 9, 31, 64, 100, 257]))
            N = max(3, steps * 512 - int(rng.integers(0, 511)))
            C = int(rng.choice([1, 15, 16, 17, 127, 128, 129, 200, 300]))
            if N * C > 6_000_000:
                C = max(1, 6_000_000 // N)
            per_pass = int(rng.choice([1, 1, 1, 2, 3, 5, 6]))
            passes = int(rng.choice([1, 2, 3, 7, 8, 31, 32, 33, 70]))
            nq = per_pass * passes
            D = random_db_matrix(rng, N, C, b)
            lo = int(rng.integers(0, N // 4 + 1)) if trial % 3 == 0 else 0
            hi = N
            dtc = orc.row_wise_compress(orc.transpose(D[lo:hi]), b)
            D_dev = torch.from_numpy(np.ascontiguousarray(D[lo:hi]).view(np.int32)).cuda()
            srv = cp.Server.from_device_matrix(D_dev, hi - lo, C, b, device=device, slot_offset=lo, total_slots=N, stream=stream)
            distinct = min(nq, 7)
            Q = np.stack([random_query(rng, N) for _ in range(distinct)])
            want = _responses(orc, Q[:, lo:hi], dtc, hi - lo, b)
            pick = np.arange(nq) % distinct
            Q_dev = torch.from_numpy(Q[pick].view(np.int32)).cuda()
            for order in (0, 1):
                cp.tuning_set("respond.interleave_passes", order)
                cp.tuning_set("respond.batch_fusion", 0 if per_pass == 1 else 1)
                R = torch.full((nq, C), -1, dtype=torch.int32, device="cuda")
                # (unfused: nq passes of one query; fused: the library cuts the nq queries into passes of its own width, up to 24)
                srv.respond_batch_device(Q_dev, nq, R, stream=stream)
                torch.cuda.synchronize()
                assert np.array_equal(R.cpu().numpy().view(np.uint32), want[pick]), (trial, b, N, C, lo, per_pass, passes, order)
            srv.close()
    finally:
        cp.tuning_reset()
