"""GPU parity at BASELINE.json's FULL size (configs[1]: 2^20 keys x 1 kB, 3-wise filter => N = 1 179 648, C = 940, b = 9)
through size-independent properties, plus one direct comparison with the oracle.

The small-size suites (test_gpu_respond.py, test_gpu_setup.py) compare every output with the oracle; at this size the
oracle needs seconds per matrix product, so most checks here are algebraic identities that only hold when every field of
the 1.5 GB packed database is read exactly once, at the right slot, with wrap-around u32 arithmetic:

  * unit queries k*e_n read back k*D[n][:] (the row regenerated on the host from the counter-based generator),
  * linearity of respond over Z/2^32,
  * exact 64-bit reference sums computed by torch from the UNPACKED matrix (independent of packing, kernel and oracle),
  * shard partials sum to the whole (the multi-GPU exchange step),
  * fused batches equal single responds,
  * all three packings (reference words, dense64, planar = the matrix-core path) answer identically,
  * the hint obeys  respond(w^T A) = w^T hint  for random w (a Freivalds check of all 1774 x 940 x 1 179 648 MACs),
  * README byte sizes of the reference (hint 6 670 248 B, query 4 718 600 B, response 3 768 B).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_KEYS, ARITY, VALUE_BYTES = 1 << 20, 3, 1024
PACKINGS = ["reference", "dense64", "planar"]
SEED_D = 0xD
SEED_MU = bytes(range(32))


class Full:
    pass


@pytest.fixture(scope="module")
def full(native, device, orc):
    """the full-size synthetic encoded DB in HBM (unpacked, 4.4 GB) and one server per packing built from it"""
    import torch

    import chalametpir_amd as cp

    f = Full()
    f.b = cp.find_encoded_db_matrix_element_bit_length(N_KEYS)
    _, _, f.N = cp.filter_shape(ARITY, N_KEYS)
    f.C = cp.encoded_num_cols(VALUE_BYTES, f.b)
    assert (f.N, f.C, f.b) == (1179648, 940, 9)  # pinned by the reference README's byte sizes (README.md:33-36)
    f.mask = (1 << f.b) - 1
    f.stream = torch.cuda.current_stream()
    f.D = torch.empty((f.N, f.C), dtype=torch.int32, device="cuda")
    device.synth_fill(f.D, f.N * f.C, SEED_D, mask=f.mask, stream=f.stream)
    f.servers = {}
    for name, dense, planar in (("reference", 0, 0), ("dense64", 1, 0), ("planar", 1, 1)):
        cp.tuning_set("layout.dense", dense)
        cp.tuning_set("layout.planar", planar)
        f.servers[name] = cp.Server.from_device_matrix(f.D, f.N, f.C, f.b, device=device, stream=f.stream)
    cp.tuning_set("layout.dense", 1)
    cp.tuning_set("layout.planar", 1)
    assert [f.servers[k].layout.packing for k in ("reference", "dense64", "planar")] == [0, 1, 2]
    torch.cuda.synchronize()
    yield f
    for s in f.servers.values():
        s.close()
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


def exact_reference_response(f, q_dev):
    """sum_n q[n] * D[n][c] in 64-bit integers on the unpacked matrix (q < 2^32, D < 2^9, N < 2^21: no overflow), low 32 bits"""
    import torch

    acc = torch.zeros(f.C, dtype=torch.int64, device="cuda")
    step = 1 << 17
    for lo in range(0, f.N, step):
        qq = q_dev[lo:lo + step].to(torch.int64) & 0xFFFFFFFF
        acc += (qq[:, None] * f.D[lo:lo + step].to(torch.int64)).sum(dim=0)
    return (acc & 0xFFFFFFFF).cpu().numpy().astype(np.uint32)


@pytest.mark.parametrize("packing", PACKINGS)
def test_unit_queries_read_back_database_rows(packing, full, device, orc):
    import torch

    f, srv = full, full.servers[packing]
    spc = int(srv.layout.slots_per_chunk)
    rng = np.random.default_rng(5)
    slots = [0, 1, 2, 3, 5, 6, 7, spc - 1, spc, spc + 1, 1023, 1024, 4095, 4096, f.N // 2, f.N - spc, f.N - 2, f.N - 1]
    slots += [int(x) for x in rng.integers(0, f.N, size=14)]
    q = torch.zeros(f.N, dtype=torch.int32, device="cuda")
    for n in slots:
        k = int(rng.integers(1, 1 << 32))
        q.zero_()
        q[n] = k - (1 << 32) if k >= (1 << 31) else k
        want = (orc.synth_fill_u32(f.C, SEED_D, n * f.C, f.mask).astype(np.uint64) * k).astype(np.uint32)
        assert np.array_equal(respond(f, srv, q), want), n


@pytest.mark.parametrize("packing", PACKINGS)
def test_random_and_all_ones_queries_match_exact_64bit_sums(packing, full, device):
    import torch

    f, srv = full, full.servers[packing]
    for seed in (0x1000, 0x1001):
        q = synth_query(f, device, seed)
        assert np.array_equal(respond(f, srv, q), exact_reference_response(f, q)), seed
    ones = torch.ones(f.N, dtype=torch.int32, device="cuda")
    col_sums = (f.D.sum(dim=0, dtype=torch.int64) & 0xFFFFFFFF).cpu().numpy().astype(np.uint32)
    assert np.array_equal(respond(f, srv, ones), col_sums)
    top = torch.full((f.N,), -1, dtype=torch.int32, device="cuda")  # q = 2^32 - 1 everywhere: r = -column sums mod 2^32
    assert np.array_equal(respond(f, srv, top), (0 - col_sums.astype(np.int64)).astype(np.uint32))


@pytest.mark.parametrize("packing", PACKINGS)
def test_respond_is_linear_mod_2_32(packing, full, device):
    f, srv = full, full.servers[packing]
    q1, q2 = synth_query(f, device, 0x2001), synth_query(f, device, 0x2002)
    r1, r2 = respond(f, srv, q1), respond(f, srv, q2)
    assert np.array_equal(respond(f, srv, q1 + q2), r1 + r2)  # int32 tensor add and uint32 numpy add both wrap
    assert np.array_equal(respond(f, srv, q1 * 3 - q2), r1 * np.uint32(3) - r2)


def test_both_packings_and_the_oracle_agree_on_a_full_size_query(full, device, orc):
    f = full
    dtc = f.servers["dense64"].export_compressed()
    assert dtc.shape == (940, 393216)
    assert np.array_equal(dtc, f.servers["reference"].export_compressed())
    assert np.array_equal(dtc, f.servers["planar"].export_compressed())
    # spot-check the packed words against the generator: row c, word w holds slots 3w..3w+2 in 10-bit lanes (matrix.rs:131-147)
    rng = np.random.default_rng(9)
    for c, w in zip(rng.integers(0, f.C, size=64), rng.integers(0, dtc.shape[1], size=64)):
        fields = [int(orc.synth_fill_u32(1, SEED_D, (3 * int(w) + j) * f.C + int(c), f.mask)[0]) for j in range(3)]
        assert int(dtc[c, w]) == fields[0] | (fields[1] << 10) | (fields[2] << 20)
    q = synth_query(f, device, 0x3003)
    q_host = q.cpu().numpy().view(np.uint32)
    want = orc.row_vector_x_compressed_transposed_matrix(q_host, dtc, f.N, f.b)[0]
    for srv in f.servers.values():
        assert np.array_equal(respond(f, srv, q), want)
    # wire sizes of the reference README: query 4 718 600 B in, response 3 768 B out, through Server::respond on host bytes
    query = np.array([1, f.N], dtype="<u4").tobytes() + q_host.tobytes()
    assert len(query) == 4718600
    want_bytes = np.array([1, f.C], dtype="<u4").tobytes() + want.tobytes()
    for name in PACKINGS:
        resp = f.servers[name].respond(query)
        assert len(resp) == 3768 and resp == want_bytes, name
    import chalametpir_amd as cp

    pin = cp.PinnedArray(f.N)  # page-locked query: DMA straight from the caller's buffer
    pin.array[:] = q_host
    assert np.array_equal(f.servers["planar"].respond_array(pin.array), want)
    pin.close()


@pytest.mark.parametrize("world", [2, 8])
def test_shard_partials_sum_to_the_whole(world, full, device):
    """the multi-GPU exchange step at full size: N split `world` ways, partial responses summed with wrap-around"""
    import torch

    import chalametpir_amd as cp
    from chalametpir_amd.distributed import shard_range

    f = full
    q = synth_query(f, device, 0x4004)
    want = respond(f, f.servers["dense64"], q)
    unit = int(f.servers["dense64"].layout.slots_per_chunk)
    total = torch.zeros(f.C, dtype=torch.int32, device="cuda")
    covered = 0
    for rank in range(world):
        lo, hi = shard_range(f.N, unit, rank, world)
        covered += hi - lo
        srv = cp.Server.from_device_matrix(f.D[lo:hi], hi - lo, f.C, f.b, device=device, slot_offset=lo, total_slots=f.N, stream=f.stream)
        part = torch.empty(f.C, dtype=torch.int32, device="cuda")
        srv.respond_device(q, part, stream=f.stream)
        total += part
        torch.cuda.synchronize()
        srv.close()
    assert covered == f.N
    assert np.array_equal(total.cpu().numpy().view(np.uint32), want)


@pytest.mark.parametrize("world", [2, 8])
def test_a_step_of_the_multi_gpu_bench_shard_by_shard(world, full, device):
    """what `bench.py --gpus N` does in one step, rank after rank on the one GPU at FULL size: every shard (boundaries as the bench cuts
    them: multiples of the planar shard unit) answers the step's 32 queries as 32 independent passes of ONE launch -- at these shard sizes
    in the interleaved order -- and as fused passes; the partial responses are summed with wrap-around (what the all-reduce does) and must
    equal the whole database's responses, two of which are held against exact 64-bit sums"""
    import torch

    import chalametpir_amd as cp
    from chalametpir_amd.distributed import shard_range

    f = full
    nq = 32
    Q = torch.empty((nq, f.N), dtype=torch.int32, device="cuda")
    for i in range(nq):
        device.synth_fill(Q, f.N, 0x6100 + i, offset_words=i * f.N, stream=f.stream)
    whole = f.servers["planar"]
    want = torch.empty((nq, f.C), dtype=torch.int32, device="cuda")
    whole.respond_batch_device(Q, nq, want, stream=f.stream)
    torch.cuda.synchronize()
    want = want.cpu().numpy().view(np.uint32)
    for i in (0, nq - 1):
        assert np.array_equal(want[i], exact_reference_response(f, Q[i])), i
    layout = cp.dtc_layout_for(f.N, f.C, f.b)
    assert int(layout.packing) == 2
    try:
        for fusion in (0, 1):
            cp.tuning_set("respond.batch_fusion", fusion)
            total = torch.zeros((nq, f.C), dtype=torch.int32, device="cuda")
            covered = 0
            for rank in range(world):
                lo, hi = shard_range(f.N, layout, rank, world)
                covered += hi - lo
                srv = cp.Server.from_device_matrix(f.D[lo:hi], hi - lo, f.C, f.b, device=device, slot_offset=lo, total_slots=f.N, stream=f.stream)
                part = torch.full((nq, f.C), -1, dtype=torch.int32, device="cuda")
                srv.respond_batch_device(Q, nq, part, stream=f.stream)
                total += part
                torch.cuda.synchronize()
                srv.close()
            assert covered == f.N
            assert np.array_equal(total.cpu().numpy().view(np.uint32), want), (world, fusion)
    finally:
        cp.tuning_reset()


@pytest.mark.parametrize("packing", PACKINGS)
def test_batches_equal_single_responds(packing, full, device):
    import torch

    import chalametpir_amd as cp

    f, srv = full, full.servers[packing]
    batch = 6
    Q = torch.empty((batch, f.N), dtype=torch.int32, device="cuda")
    for i in range(batch):
        device.synth_fill(Q, f.N, 0x5000 + i, offset_words=i * f.N, stream=f.stream)
    singles = np.stack([respond(f, srv, Q[i]) for i in range(batch)])
    try:
        for fusion in (1, 0):
            cp.tuning_set("respond.batch_fusion", fusion)
            R = torch.full((batch, f.C), -1, dtype=torch.int32, device="cuda")
            srv.respond_batch_device(Q, batch, R, stream=f.stream)
            torch.cuda.synchronize()
            assert np.array_equal(R.cpu().numpy().view(np.uint32), singles), fusion
    finally:
        cp.tuning_set("respond.batch_fusion", 1)


def test_full_size_setup_hint_passes_freivalds_check(full, device, orc):
    """Server::setup's matrix half at full size (server.rs:59-67): hint = A*D with A = generate_from_seed(1774, N, seed).
    For random w in (Z/2^32)^1774:  (w^T A) * D = w^T (A*D), so respond(w^T A) must equal w^T hint; a single wrong hint entry
    survives one random w with probability <= 1/2 per low bit affected, so several w are used.  The first rows of A (and hence of
    the hint) are also compared entry for entry with the oracle's XOF and exact 64-bit sums."""
    import torch

    import chalametpir_amd as cp

    f = full
    D_host = f.D.cpu().numpy().view(np.uint32)
    srv, hint = cp.Server.setup_from_matrix(SEED_MU, D_host, f.b, device=device)
    del D_host
    assert hint.shape == (1774, f.C) and 8 + hint.nbytes == 6670248  # README.md:33
    ph = srv.setup_timings()
    assert ph["total"] > 0 and ph["hint_matmul"] > 0
    # the server built by setup holds the same packed DB as the one packed from the device matrix
    q = synth_query(f, device, 0x6006)
    assert np.array_equal(respond(f, srv, q), respond(f, f.servers["dense64"], q))

    A = cp.generate_from_seed(1774, f.N, SEED_MU)  # product XOF (host); its prefix is checked against the oracle's below
    rows = 3
    assert np.array_equal(A[:rows], orc.generate_from_seed(rows, f.N, SEED_MU))
    for r in range(rows):
        a = torch.from_numpy(A[r].view(np.int32)).cuda()
        assert np.array_equal(hint[r], exact_reference_response(f, a)), r

    rng = np.random.default_rng(77)
    n_w = 4
    Wt = rng.integers(0, 1 << 32, size=(n_w, 1774), dtype=np.uint64)
    wA = torch.zeros((n_w, f.N), dtype=torch.int64, device="cuda")
    blk = 128
    for r0 in range(0, 1774, blk):
        a = torch.from_numpy(A[r0:r0 + blk].view(np.int32)).cuda().to(torch.int64) & 0xFFFFFFFF
        w = torch.from_numpy(Wt[:, r0:r0 + blk].astype(np.int64)).cuda()
        for i in range(n_w):  # int64 products wrap mod 2^64; only the low 32 bits are used
            wA[i] += (w[i][:, None] * a).sum(dim=0)
    del A
    hint64 = hint.astype(np.uint64)
    for i in range(n_w):
        lhs = respond(f, srv, _low32_as_int32(wA[i]))
        rhs = ((Wt[i][:, None] * hint64).sum(axis=0, dtype=np.uint64) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        assert np.array_equal(lhs, rhs), i
    srv.close()


def _low32_as_int32(t):
    """low 32 bits of an int64 tensor, reinterpreted as int32 (two's complement)"""
    import torch

    lo = t & 0xFFFFFFFF
    lo = torch.where(lo >= (1 << 31), lo - (1 << 32), lo)
    return lo.to(torch.int32).contiguous()
