"""GPU parity at the shapes of EVERY remaining BASELINE.json config, plus the rest of the reference's bench grid
(integrations/benches/online_phase.rs:40-57: 2^16 / 2^18 / 2^20 keys x arity {3, 4}; configs[1] = 2^20 x 3-wise has its own module,
test_gpu_fullsize.py):

    configs[2]  2^20 keys, 1 kB values, 4-wise filter   N = 1 130 496, C = 940            (query 4 521 992 B: reference README.md:35)
    configs[3]  2^22 keys, 1 kB values, 3-wise          N = 4 718 592, C = 940            N*C and 1774*N exceed 2^32
    configs[4]  2^20 keys, 8 kB values, 3-wise          N = 1 179 648, C = 7 312          N*C exceeds 2^32; 457 column tiles

The reference itself cannot run configs[3] and configs[4]: it sizes its buffers in u32 (chalametpir_common/src/matrix.rs:50,71,546,988,1048),
so there "parity" is parity with its arithmetic definition carried out with 64-bit indexing -- exactly what these tests pin down.

The encoded database is synthetic (counter-based generator) and lives only in HBM (up to 34.5 GB unpacked); checks that
need the oracle rebuild single COLUMNS of it on the host with an independent numpy statement of the generator (tests/_cases.py), pack
them with the oracle (transpose -> row_wise_compress) and run the oracle's mat-vec on them.  Everything goes through the C ABI.
"""
import numpy as np
import pytest

from _cases import synth_u32_at, wire

pytestmark = pytest.mark.gpu

SEED_D, SEED_A = 0xD, 0xA
#          name          keys     arity value  (N, C, b) expected (None = not pinned there)
CONFIGS = [
    ("2^16x3", 1 << 16, 3, 1024, (77824, 846, 10)),
    ("2^16x4", 1 << 16, 4, 1024, None),
    ("2^18x3", 1 << 18, 3, 1024, None),
    ("2^18x4", 1 << 18, 4, 1024, None),
    ("cfg3:2^20x4", 1 << 20, 4, 1024, (1130496, 940, 9)),
    ("cfg4:2^22x3", 1 << 22, 3, 1024, (4718592, 940, 9)),
    ("cfg5:2^20x3x8kB", 1 << 20, 3, 8192, (1179648, 7312, 9)),
]


class Cfg:
    pass


@pytest.fixture(scope="module", params=CONFIGS, ids=[c[0] for c in CONFIGS])
def cfg(request, native, device):
    """the config's synthetic encoded DB in HBM (unpacked) and the server packed from it (default packing: planar for b >= 9)"""
    import torch

    import chalametpir_amd as cp

    name, n_keys, arity, value_bytes, expect = request.param
    f = Cfg()
    f.name, f.n_keys, f.arity = name, n_keys, arity
    f.b = cp.find_encoded_db_matrix_element_bit_length(n_keys)
    _, _, f.N = cp.filter_shape(arity, n_keys)
    f.C = cp.encoded_num_cols(value_bytes, f.b)
    if expect is not None:
        assert (f.N, f.C, f.b) == expect
    f.cf = 2 if f.b >= 11 else (3 if f.b >= 9 else 4)
    f.mask = (1 << f.b) - 1
    f.stream = torch.cuda.current_stream()
    f.D = torch.empty((f.N, f.C), dtype=torch.int32, device="cuda")
    device.synth_fill(f.D, f.N * f.C, SEED_D, mask=f.mask, stream=f.stream)
    f.srv = cp.Server.from_device_matrix(f.D, f.N, f.C, f.b, device=device, stream=f.stream)
    torch.cuda.synchronize()
    assert f.srv.layout.packing == 2  # the matrix-core path: what every BASELINE config runs on
    yield f
    f.srv.close()
    del f.D
    torch.cuda.empty_cache()


def respond(f, srv, q_dev):
    import torch

    r = torch.empty(f.C, dtype=torch.int32, device="cuda")
    srv.respond_device(q_dev, r, stream=f.stream)
    torch.cuda.synchronize()
    return r.cpu().numpy().view(np.uint32)


def synth_query(f, device, seed):
    import torch

    q = torch.empty(f.N, dtype=torch.int32, device="cuda")
    device.synth_fill(q, f.N, seed, stream=f.stream)
    return q


def exact_sums(f, q_dev, D=None):
    """sum_n q[n] * D[n][c] in 64-bit integers on the UNPACKED matrix (q < 2^32, D < 2^14, N < 2^23: no overflow), low 32 bits"""
    import torch

    D = f.D if D is None else D
    acc = torch.zeros(D.shape[1], dtype=torch.int64, device="cuda")
    step = max(1024, (1 << 27) // D.shape[1])  # ~1 GiB of int64 products per slice
    for lo in range(0, D.shape[0], step):
        qq = q_dev[lo:lo + step].to(torch.int64) & 0xFFFFFFFF
        acc += (qq[:, None] * D[lo:lo + step].to(torch.int64)).sum(dim=0)
    return (acc & 0xFFFFFFFF).cpu().numpy().astype(np.uint32)


def host_columns(f, cols):
    """columns `cols` of the synthetic D rebuilt on the host, independent of the device generator: N x len(cols) u32"""
    n = np.arange(f.N, dtype=np.uint64)[:, None] * np.uint64(f.C)
    return synth_u32_at(n + np.asarray(cols, dtype=np.uint64)[None, :], SEED_D, f.mask)


def test_unit_queries_read_back_database_rows(cfg, device):
    import torch

    f = cfg
    rng = np.random.default_rng(5)
    slots = [0, 1, 2, 3, 63, 64, 511, 512, 513, 4095, 4096, f.N // 2, f.N - 513, f.N - 512, f.N - 2, f.N - 1]
    slots += [int(x) for x in rng.integers(0, f.N, size=8)]
    q = torch.zeros(f.N, dtype=torch.int32, device="cuda")
    for n in slots:
        k = int(rng.integers(1, 1 << 32))
        q.zero_()
        q[n] = k - (1 << 32) if k >= (1 << 31) else k
        row = synth_u32_at(np.uint64(n) * np.uint64(f.C) + np.arange(f.C, dtype=np.uint64), SEED_D, f.mask)
        want = (row.astype(np.uint64) * np.uint64(k)).astype(np.uint32)
        assert np.array_equal(respond(f, f.srv, q), want), n


def test_random_and_extreme_queries_match_exact_64bit_sums(cfg, device):
    import torch

    f = cfg
    for seed in (0x1000, 0x1001):
        q = synth_query(f, device, seed)
        assert np.array_equal(respond(f, f.srv, q), exact_sums(f, q)), seed
    col_sums = np.zeros(f.C, dtype=np.uint64)
    step = max(1024, (1 << 28) // f.C)
    for lo in range(0, f.N, step):
        col_sums += f.D[lo:lo + step].sum(dim=0, dtype=torch.int64).cpu().numpy().astype(np.uint64)
    col_sums = (col_sums & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    ones = torch.ones(f.N, dtype=torch.int32, device="cuda")  # the reference's own property: all-ones query = column sums (matrix.rs:1319-1376)
    assert np.array_equal(respond(f, f.srv, ones), col_sums)
    top = torch.full((f.N,), -1, dtype=torch.int32, device="cuda")  # q = 2^32 - 1 everywhere: r = -column sums mod 2^32
    assert np.array_equal(respond(f, f.srv, top), (0 - col_sums.astype(np.int64)).astype(np.uint32))


def test_respond_is_linear_mod_2_32(cfg, device):
    f = cfg
    q1, q2 = synth_query(f, device, 0x2001), synth_query(f, device, 0x2002)
    r1, r2 = respond(f, f.srv, q1), respond(f, f.srv, q2)
    assert np.array_equal(respond(f, f.srv, q1 + q2), r1 + r2)  # int32 tensor add and uint32 numpy add both wrap
    assert np.array_equal(respond(f, f.srv, q1 * 3 - q2), r1 * np.uint32(3) - r2)


def test_oracle_on_host_rebuilt_columns_and_wire_bytes(cfg, device, orc):
    """the oracle's transpose -> row_wise_compress -> mat-vec (matrix.rs:517-527, 98-205, 328-485) on columns of D rebuilt on the host
    must give the same response words as the device; then Server::respond on wire bytes (server.rs:184-190) with the reference's sizes"""
    f = cfg
    rng = np.random.default_rng(11)
    last_tile = (f.C - 1) // 16 * 16
    cols = sorted({0, 1, 15, 16, 17, 63, 64, f.C // 2, last_tile - 1, last_tile, f.C - 2, f.C - 1} | {int(c) for c in rng.integers(0, f.C, size=20)})
    D_sub = host_columns(f, cols)
    dtc_sub = orc.row_wise_compress(orc.transpose(D_sub), f.b)
    assert dtc_sub.shape == (len(cols), -(-f.N // f.cf))
    q = synth_query(f, device, 0x3003)
    q_host = q.cpu().numpy().view(np.uint32)
    assert np.array_equal(q_host[:4096], synth_u32_at(np.arange(4096), 0x3003))
    want = orc.row_vector_x_compressed_transposed_matrix(q_host, dtc_sub, f.N, f.b)[0]
    got = respond(f, f.srv, q)
    assert np.array_equal(got[cols], want)
    # wire images: query 8 + 4N bytes in, response 8 + 4C bytes out (matrix.rs:947-1010); pageable and page-locked host buffers
    query = wire(q_host)
    assert len(query) == 8 + 4 * f.N
    if f.name.startswith("cfg3"):
        assert len(query) == 4521992  # reference README.md:35 (4-wise filter, 2^20 keys)
    resp = f.srv.respond(query)
    assert resp == wire(got) and len(resp) == 8 + 4 * f.C
    import chalametpir_amd as cp

    pin = cp.PinnedArray(f.N)
    pin.array[:] = q_host
    assert np.array_equal(f.srv.respond_array(pin.array), got)
    pin.close()
    # malformed queries fail as Matrix::from_bytes / the 1 x N check do (matrix.rs:973-1010, 329-331)
    with pytest.raises(cp.ChalametPIRError) as e:
        f.srv.respond(query[:-4])
    assert e.value.code == 7
    with pytest.raises(cp.ChalametPIRError) as e:
        f.srv.respond(wire(q_host[:-1]))
    assert e.value.code == 5


def test_packed_database_exports_to_the_reference_representation(cfg, orc):
    """compressed_transposed_parsed_db_mat_d as the reference holds it (server.rs:18), up to 11.5 GB on the host (configs[4])"""
    f = cfg
    words = f.C * -(-f.N // f.cf)
    if words * 4 > (16 << 30):
        pytest.skip("more than 16 GB on the host")
    dtc = f.srv.export_compressed()
    rng = np.random.default_rng(13)
    cols = sorted({0, f.C - 1} | {int(c) for c in rng.integers(0, f.C, size=6)})
    want = orc.row_wise_compress(orc.transpose(host_columns(f, cols)), f.b)
    assert np.array_equal(dtc[cols], want)


def test_shard_partials_sum_to_the_whole_8_ways(cfg, device):
    """BASELINE's 8-GPU placement: N split 8 ways (the shapes each GPU really holds: 739 MB of the reference packing per shard at
    configs[3], 1.44 GB at configs[4]), partial responses summed with wrap-around = the RCCL reduce of SURVEY.md 8e"""
    import torch

    import chalametpir_amd as cp
    from chalametpir_amd.distributed import shard_range

    f = cfg
    world = 8
    q = synth_query(f, device, 0x4004)
    want = respond(f, f.srv, q)
    total = torch.zeros(f.C, dtype=torch.int32, device="cuda")
    covered, sizes = 0, []
    for rank in range(world):
        lo, hi = shard_range(f.N, f.srv.layout, rank, world)
        covered += hi - lo
        sizes.append(hi - lo)
        if hi == lo:
            continue
        assert lo % 512 == 0 and lo % f.cf == 0
        srv = cp.Server.from_device_matrix(f.D[lo:hi], hi - lo, f.C, f.b, device=device, slot_offset=lo, total_slots=f.N, stream=f.stream)
        part = torch.empty(f.C, dtype=torch.int32, device="cuda")
        srv.respond_device(q, part, stream=f.stream)
        total += part
        torch.cuda.synchronize()
        if rank in (0, world - 1):  # a shard answers a host query from ITS slice of the wire image only
            part_host = srv.respond_array(q.cpu().numpy().view(np.uint32))
            assert np.array_equal(part_host, part.cpu().numpy().view(np.uint32))
        srv.close()
    assert covered == f.N
    if f.name.startswith("cfg4"):
        assert sizes == [589824] * 8 and 4 * f.C * (sizes[0] // f.cf) == 739246080
    assert np.array_equal(total.cpu().numpy().view(np.uint32), want)


def test_batches_equal_single_responds(cfg, device):
    import torch

    import chalametpir_amd as cp

    f = cfg
    batch = 11  # ONE pass of the wide kernel (three of its row sets) when fused; 11 independent passes in one launch when not
    Q = torch.empty((batch, f.N), dtype=torch.int32, device="cuda")
    for i in range(batch):
        device.synth_fill(Q, f.N, 0x5000 + i, offset_words=i * f.N, stream=f.stream)
    singles = np.stack([respond(f, f.srv, Q[i]) for i in range(batch)])
    assert np.array_equal(singles[0], exact_sums(f, Q[0]))
    try:
        for fusion in (1, 0):
            cp.tuning_set("respond.batch_fusion", fusion)
            R = torch.full((batch, f.C), -1, dtype=torch.int32, device="cuda")
            f.srv.respond_batch_device(Q, batch, R, stream=f.stream)
            torch.cuda.synchronize()
            assert np.array_equal(R.cpu().numpy().view(np.uint32), singles), fusion
    finally:
        cp.tuning_set("respond.batch_fusion", 1)


def test_passes_of_one_launch_in_both_orders_at_full_size(cfg, device):
    """what bench.py's step is at this config -- many independent passes (one query each) in ONE launch -- in slice order (every pass its own
    stream of the database: the N = 1 headline) and in the interleaved order (the (pass, unit) space shared out: what the shards of a
    multi-GPU run dispatch, blocks straddling passes): the same responses as single launches, two of them checked against exact 64-bit
    sums; 13 passes so that the passes do not divide the blocks of an XCD evenly"""
    import torch

    import chalametpir_amd as cp

    f = cfg
    passes = 13
    Q = torch.empty((passes, f.N), dtype=torch.int32, device="cuda")
    for i in range(passes):
        device.synth_fill(Q, f.N, 0x9100 + i, offset_words=i * f.N, stream=f.stream)
    singles = np.stack([respond(f, f.srv, Q[i]) for i in range(passes)])
    assert np.array_equal(singles[0], exact_sums(f, Q[0])) and np.array_equal(singles[passes - 1], exact_sums(f, Q[passes - 1]))
    try:
        cp.tuning_set("respond.batch_fusion", 0)
        for order in (0, 1):
            cp.tuning_set("respond.interleave_passes", order)
            R = torch.full((passes, f.C), -1, dtype=torch.int32, device="cuda")
            f.srv.respond_batch_device(Q, passes, R, stream=f.stream)
            torch.cuda.synchronize()
            assert np.array_equal(R.cpu().numpy().view(np.uint32), singles), order
    finally:
        cp.tuning_reset()


def test_step_major_kernel_everywhere_equals_the_default(cfg, device):
    """respond.ks_major=2: the step-major kernel (every query word read once; the lone host caller's kernel) answers the device-resident
    queries too, fused batches in passes of 4; 3: the same in the strided step order of the in-place host path.  Same responses as the wide
    kernel, which takes every device launch by default."""
    import torch

    import chalametpir_amd as cp

    f = cfg
    batch = 5
    Q = torch.empty((batch, f.N), dtype=torch.int32, device="cuda")
    for i in range(batch):
        device.synth_fill(Q, f.N, 0x7100 + i, offset_words=i * f.N, stream=f.stream)
    want = np.stack([respond(f, f.srv, Q[i]) for i in range(batch)])
    assert np.array_equal(want[1], exact_sums(f, Q[1]))
    try:
        for mode in (2, 3):
            cp.tuning_set("respond.ks_major", mode)
            got = np.stack([respond(f, f.srv, Q[i]) for i in range(batch)])
            assert np.array_equal(got, want), mode
            # (3 = as the in-place host path launches it: whole steps round-robin over the blocks, the left-over steps split by units,
            # fragments built a visit ahead; a pass whose responses exceed its LDS accumulators goes to the wide kernel)
            R = torch.full((batch, f.C), -1, dtype=torch.int32, device="cuda")
            f.srv.respond_batch_device(Q, batch, R, stream=f.stream)
            torch.cuda.synchronize()
            assert np.array_equal(R.cpu().numpy().view(np.uint32), want), mode
    finally:
        cp.tuning_set("respond.ks_major", 1)


def test_hint_matmul_at_this_shape(cfg, device):
    """impl Mul for &Matrix (matrix.rs:1040-1059) == gpu_utils::mat_x_mat (gpu_utils.rs:156-220) at the config's shape: hint = A * D with
    a synthetic A (the XOF expansion of the real A is a host phase, covered at configs[1] in test_gpu_fullsize.py).  Checked two ways:
    exact 64-bit sums of the first and last rows, and Freivalds: respond(w^T A) = w^T hint for random w in (Z/2^32)^1774."""
    import torch

    f = cfg
    R = 1774
    A = torch.empty((R, f.N), dtype=torch.int32, device="cuda")
    device.synth_fill(A, R * f.N, SEED_A, stream=f.stream)
    M = torch.full((R, f.C), -1, dtype=torch.int32, device="cuda")
    device.mat_x_mat(A, f.D, M, R, f.N, f.C, rhs_max_bits=16, stream=f.stream)
    torch.cuda.synchronize()
    hint = M.cpu().numpy().view(np.uint32)
    for r in (0, 1, 127, 128, R - 1):
        assert np.array_equal(hint[r], exact_sums(f, A[r])), r
    # the same product with the right-hand side taken from the packed image + the plane written beside it (what Server::setup runs for
    # b >= 9): same hint, bit for bit, and the same image
    import chalametpir_amd as cp

    L = f.srv.layout
    plane_bytes = cp.packed_rhs_plane_bytes(L)  # 0 with b = 9: the matmul expands the image's one bit plane itself
    if cp.packed_rhs_offered(L):
        dtc = torch.empty(int(L.total_words), dtype=torch.int32, device="cuda")
        plane = torch.empty(plane_bytes // 4, dtype=torch.int32, device="cuda") if plane_bytes else None
        flag = torch.zeros(1, dtype=torch.int32, device="cuda")
        device.transpose_compress_with_plane(f.D, L, dtc, plane, or_of_entries=flag, stream=f.stream)
        M2 = torch.full((R, f.C), -1, dtype=torch.int32, device="cuda")
        device.mat_x_packed(A, dtc, L, plane, M2, R, stream=f.stream)
        torch.cuda.synchronize()
        assert int(flag.item()) >> f.b == 0
        assert torch.equal(M2, M)
        del dtc, plane, M2
    rng = np.random.default_rng(77)
    n_w = 3
    Wt = rng.integers(0, 1 << 32, size=(n_w, R), dtype=np.uint64)
    wA = torch.zeros((n_w, f.N), dtype=torch.int64, device="cuda")
    blk = max(1, (1 << 27) // f.N)  # ~1 GiB of int64 per slice of A
    for r0 in range(0, R, blk):
        a = A[r0:r0 + blk].to(torch.int64) & 0xFFFFFFFF
        w = torch.from_numpy(Wt[:, r0:r0 + blk].astype(np.int64)).cuda()
        for i in range(n_w):  # int64 products wrap mod 2^64; only the low 32 bits are used
            wA[i] += (w[i][:, None] * a).sum(dim=0)
        del a
    del A
    hint64 = hint.astype(np.uint64)
    for i in range(n_w):
        lo = wA[i] & 0xFFFFFFFF
        lo = torch.where(lo >= (1 << 31), lo - (1 << 32), lo).to(torch.int32).contiguous()
        lhs = respond(f, f.srv, lo)
        rhs = ((Wt[i][:, None] * hint64).sum(axis=0, dtype=np.uint64) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        assert np.array_equal(lhs, rhs), i
    torch.cuda.empty_cache()
