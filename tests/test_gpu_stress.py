"""Life-cycle stress of the HOST entry point (cpir_server_respond), part of `-m gpu` (what scripts/lifecycle_soak.py and scripts/soak.py
do for minutes, here for a bounded number of rounds): servers of three shapes are created, asked by

  * a lone caller with a PAGEABLE query (2^19+ words: one launch polling the copy's progress; below: staged, read in place from the
    arena's pinned block),
  * a lone caller whose query lies in page-locked memory (cpir_host_alloc: read in place from the caller's buffer),
  * a lone caller whose buffer (a mapping of its own, never heap memory) is hipHostRegister'ed, asked, unregistered and asked again at
    the SAME address (must be staged then),
  * bursts of concurrent callers mixing pageable and page-locked buffers (arenas coalesce and pipeline them),

cloned, closed while the clone keeps answering, and destroyed -- interleaved, so that arenas, streams, pinned blocks and registrations are
created and torn down next to each other.  Every response is compared with the oracle.  Models the reference's serving loop: one
Arc<Server> answered from many tasks, servers replaced while others serve (chalametpir_server/examples/server.rs:45-93)."""
import threading
import time

import numpy as np
import pytest

from _cases import OwnMapping, random_db_matrix, random_query

pytestmark = pytest.mark.gpu

SHAPES = [((1 << 19) + 4096 * 3 + 5, 24, 9), (77_824, 130, 10), (600_000, 7, 12), (3 * 1536 + 1, 19, 6)]


@pytest.mark.parametrize("order", ["forward", "reversed"])
def test_server_lifecycle_under_mixed_callers(order, orc, device):
    import torch

    import chalametpir_amd as cp

    rt = torch.cuda.cudart()
    can_register = hasattr(rt, "cudaHostRegister") and hasattr(rt, "cudaHostUnregister")
    rng = np.random.default_rng(31337)
    shapes = SHAPES if order == "forward" else SHAPES[::-1]
    t_end = time.time() + 25  # a bound, not a target: three rounds over the shapes normally take well under that
    responses = 0
    for rnd in range(3):
        for N, C, b in shapes:
            if time.time() > t_end:
                break
            D = random_db_matrix(rng, N, C, b)
            if rnd == 1:  # the middle round: a database with empty rows, served through the slot map (compact.hip) -- the lone pageable
                D[rng.random(N) < 0.12] = 0  # caller's copy jobs compact, concurrent callers are gathered on the device
            dtc = orc.row_wise_compress(orc.transpose(D), b)
            if rnd == 1:
                srv = cp.Server.from_device_matrix(torch.from_numpy(D.view(np.int32)).cuda(), N, C, b, device=device, stream=torch.cuda.current_stream())
                assert srv.slots_served()[0] < N
            else:
                srv = cp.Server.from_compressed(dtc, N, b, device=device)

            def want(q):
                return orc.row_vector_x_compressed_transposed_matrix(q, dtc, N, b)[0]

            bad = []

            def ask(server, buf, q, tag):
                if not np.array_equal(server.respond_array(buf), want(q)):
                    bad.append((rnd, N, C, b, tag))

            # lone callers: pageable, page-locked, registered -> unregistered at the same address
            q = random_query(rng, N)
            ask(srv, q, q, "pageable")
            pin = cp.PinnedArray(N)
            pin.array[:] = q
            ask(srv, pin.array, q, "page-locked")
            if can_register and rnd == 0:  # (once per shape and order: 8 register / unregister cycles per run are enough)
                own = OwnMapping(N)  # a mapping of its own: heap memory is never registered in this suite (see OwnMapping)
                qr = own.array
                qr[:] = random_query(rng, N)
                err = rt.cudaHostRegister(own.address, own.nbytes, 0)
                assert int(err) == 0, err
                try:
                    ask(srv, qr, qr.copy(), "registered")
                finally:
                    err = rt.cudaHostUnregister(own.address)
                assert int(err) == 0, err
                ask(srv, qr, qr.copy(), "unregistered-again")
                del qr
                own.close()
            # a burst of concurrent callers on the one handle, pageable and page-locked buffers mixed
            qs = [random_query(rng, N) for _ in range(6)]
            pins = [cp.PinnedArray(N) for _ in range(2)]
            for pa, qq in zip(pins, qs):
                pa.array[:] = qq
            ts = [threading.Thread(target=ask, args=(srv, pins[i].array if i < 2 else qs[i], qs[i], f"burst{i}")) for i in range(6)]
            [t.start() for t in ts]
            [t.join() for t in ts]
            # Clone shares the database; the original goes away while the clone keeps serving (server.rs:15 #[derive(Clone)])
            clone = srv.clone()
            srv.close()
            q2 = random_query(rng, N)
            ask(clone, q2, q2, "clone-after-close")
            ask(clone, pin.array, q, "clone-page-locked")
            clone.close()
            for pa in pins:
                pa.close()
            pin.close()
            assert not bad, bad
            responses += 12 if (can_register and rnd == 0) else 10
    assert responses >= 10 * len(SHAPES)  # at least one full round fitted the bound


def test_polled_and_plain_lone_launches_agree_while_servers_come_and_go(orc, device):
    """the polled launch (kernel started in front of the copy of a lone pageable query) against the plain paths on the same queries, with
    a second server being created, asked and destroyed between the calls: same answers whichever path served them"""
    import chalametpir_amd as cp

    rng = np.random.default_rng(99)
    b, N, C = 9, (1 << 19) + 2048 + 7, 9
    D = random_db_matrix(rng, N, C, b)
    dtc = orc.row_wise_compress(orc.transpose(D), b)
    qs = [random_query(rng, N) for _ in range(4)]
    wants = [orc.row_vector_x_compressed_transposed_matrix(q, dtc, N, b)[0] for q in qs]
    small = (4 * 1024 + 3, 5, 10)
    Ds = random_db_matrix(rng, *small)
    dtcs = orc.row_wise_compress(orc.transpose(Ds), small[2])
    qsmall = random_query(rng, small[0])
    wsmall = orc.row_vector_x_compressed_transposed_matrix(qsmall, dtcs, small[0], small[2])[0]
    srv = cp.Server.from_compressed(dtc, N, b, device=device)
    for timeout_us, zero_copy in ((20000, 1), (0, 1), (20000, 0), (2000, 1)):
        cp.tuning_set("respond.host_fill_timeout_us", timeout_us)
        cp.tuning_set("respond.host_zero_copy", zero_copy)
        for q, w in zip(qs, wants):
            assert np.array_equal(srv.respond_array(q), w), (timeout_us, zero_copy)
            other = cp.Server.from_compressed(dtcs, small[0], small[2], device=device)
            assert np.array_equal(other.respond_array(qsmall), wsmall)
            other.close()
    srv.close()


def test_mat_x_packed_beyond_the_pipelined_kernels_reach_is_refused_up_front(device):
    """N >= 2^23 slots: the hand-pipelined matmul cannot address a 128-row tile of A with 32-bit byte offsets.  cpir_op_mat_x_packed must
    say so BEFORE touching M (mfma_planar_rhs_applicable is what Server::setup asks to choose its path), and cpir_op_mat_x_mat still
    multiplies that shape -- matrix cores and VALU agree, spot-checked against numpy on rebuilt entries."""
    import torch

    import chalametpir_amd as cp
    from _cases import synth_u32_at

    stream = torch.cuda.current_stream()
    b, rows, N, C = 9, 3, (1 << 23) + 512, 16
    L = cp.dtc_layout_for(N, C, b, packing=2)
    D = torch.empty((N, C), dtype=torch.int32, device="cuda")
    A = torch.empty((rows, N), dtype=torch.int32, device="cuda")
    device.synth_fill(D, N * C, 0xD0, mask=(1 << b) - 1, stream=stream)
    device.synth_fill(A, rows * N, 0xA0, stream=stream)
    dtc = torch.empty(L.total_words, dtype=torch.int32, device="cuda")
    assert cp.packed_rhs_offered(L) and cp.packed_rhs_plane_bytes(L) == 0  # b = 9: the high byte comes out of the image's own bit plane
    plane = None
    device.transpose_compress_with_plane(D, L, dtc, plane, stream=stream)
    M = torch.full((rows, C), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
    with pytest.raises(cp.ChalametPIRError):
        device.mat_x_packed(A, dtc, L, plane, M, rows, stream=stream)
    torch.cuda.synchronize()
    assert bool((M == 0x5A5A5A5A).all())  # refused before M was zeroed
    got = {}
    for mfma in (1, 0):
        cp.tuning_set("matmul.mfma", mfma)
        Mx = torch.empty((rows, C), dtype=torch.int32, device="cuda")
        device.mat_x_mat(A, D, Mx, rows, N, C, rhs_max_bits=16, stream=stream)
        torch.cuda.synchronize()
        got[mfma] = Mx.cpu().numpy().view(np.uint32)
    assert np.array_equal(got[0], got[1])
    n = np.arange(N, dtype=np.uint64)
    for r, c in ((0, 0), (2, 15), (1, 7)):
        a = synth_u32_at(np.uint64(r) * np.uint64(N) + n, 0xA0).astype(np.uint64)
        d = synth_u32_at(n * np.uint64(C) + np.uint64(c), 0xD0, (1 << b) - 1).astype(np.uint64)
        assert int((a * d).sum(dtype=np.uint64) & np.uint64(0xFFFFFFFF)) == int(got[1][r, c])


def test_scratch_blocks_are_reused_only_behind_their_events_and_refused_under_capture(orc, device):
    """The scratch pool (host_respond.hip): cpir_op_mat_x_mat on the matrix cores takes its workspace from it.  (1) Products enqueued back to
    back WITHOUT synchronising -- every one would find the block of the one before still waiting for its event -- and then again after a
    pause (the block is idle: reused) all equal the oracle's product; (2) the same entry point on a stream that is being CAPTURED into a
    graph is refused with CPIR_ERR_INVALID_ARGUMENT before anything is enqueued -- its scratch bookkeeping cannot be replayed -- and the
    capture itself stays valid (it ends in an empty graph)."""
    import ctypes

    import torch

    import chalametpir_amd as cp

    rng = np.random.default_rng(61)
    rows, N, C, b = 70, 4096 + 64, 48, 9
    A = rng.integers(0, 1 << 32, size=(rows, N), dtype=np.uint64).astype(np.uint32)
    D = random_db_matrix(rng, N, C, b)
    want = orc.mul(A, D)
    stream = torch.cuda.Stream()
    Ad, Dd = torch.from_numpy(A.view(np.int32)).cuda(), torch.from_numpy(D.view(np.int32)).cuda()
    outs = [torch.empty((rows, C), dtype=torch.int32, device="cuda") for _ in range(6)]
    torch.cuda.synchronize()
    for rnd in range(3):
        for M in outs:
            device.mat_x_mat(Ad, Dd, M, rows, N, C, rhs_max_bits=16, stream=stream)
        stream.synchronize()
        for M in outs:
            assert np.array_equal(M.cpu().numpy().view(np.uint32), want)
            M.zero_()
        torch.cuda.synchronize()
        time.sleep(0.05 if rnd == 0 else 0.7)  # (second pause: beyond the pool's idle time -- the block is freed and allocated again)
    hip = ctypes.CDLL("libamdhip64.so")
    s_ptr = ctypes.c_void_p(stream.cuda_stream)
    assert hip.hipStreamBeginCapture(s_ptr, 2) == 0  # hipStreamCaptureModeRelaxed
    try:
        with pytest.raises(cp.ChalametPIRError) as ei:
            device.mat_x_mat(Ad, Dd, outs[0], rows, N, C, rhs_max_bits=16, stream=stream)
        assert ei.value.code == 68  # CPIR_ERR_INVALID_ARGUMENT
    finally:
        graph = ctypes.c_void_p()
        assert hip.hipStreamEndCapture(s_ptr, ctypes.byref(graph)) == 0
        if graph.value:
            hip.hipGraphDestroy(graph)
    device.mat_x_mat(Ad, Dd, outs[0], rows, N, C, rhs_max_bits=16, stream=stream)  # (and the stream works as before)
    stream.synchronize()
    assert np.array_equal(outs[0].cpu().numpy().view(np.uint32), want)


def test_plain_respond_entry_points_can_be_captured_into_a_graph_and_replayed(orc, device):
    """the device-pointer respond entry points on the planar packing enqueue a memset and kernels, nothing else -- so a caller
    with a launch-bound loop may capture them into a hipGraph.  One query and a fused batch of 5 are captured once and replayed on fresh query
    contents three times: every replay equals the oracle (the first, eager call has done the one-time attribute set-up)."""
    import ctypes

    import torch

    import chalametpir_amd as cp

    _capture_and_replay(orc, device, holes=False)


def test_respond_on_a_compacted_server_can_be_captured_too(orc, device):
    """... and so can a server that keeps only the slots that hold something, where the wide kernel applies the slot map itself (no scratch,
    no gather pass in front): a fifth of the rows of D zero, the same capture and replays"""
    _capture_and_replay(orc, device, holes=True)


def _capture_and_replay(orc, device, holes):
    import ctypes

    import torch

    import chalametpir_amd as cp

    rng = np.random.default_rng(77)
    N, C, b = 3 * 1536 + 77, 33, 9
    D = random_db_matrix(rng, N, C, b)
    if holes:
        D[rng.random(N) < 0.2] = 0
    srv, _ = cp.Server.setup_from_matrix(bytes(range(32)), D, b, device=device)
    assert (srv.slots_served()[0] < N) == holes
    dtc = orc.row_wise_compress(orc.transpose(D), b)
    stream = torch.cuda.Stream()
    nb = 5
    q_dev = torch.zeros((nb, N), dtype=torch.int32, device="cuda")
    r1 = torch.zeros(C, dtype=torch.int32, device="cuda")
    rb = torch.zeros((nb, C), dtype=torch.int32, device="cuda")
    srv.respond_device(q_dev[0], r1, stream=stream)          # eager once: module load, function attributes
    srv.respond_batch_device(q_dev, nb, rb, stream=stream)
    stream.synchronize()
    hip = ctypes.CDLL("libamdhip64.so")
    s_ptr = ctypes.c_void_p(stream.cuda_stream)
    assert hip.hipStreamBeginCapture(s_ptr, 2) == 0  # hipStreamCaptureModeRelaxed
    try:
        srv.respond_device(q_dev[0], r1, stream=stream)
        srv.respond_batch_device(q_dev, nb, rb, stream=stream)
    finally:
        graph = ctypes.c_void_p()
        assert hip.hipStreamEndCapture(s_ptr, ctypes.byref(graph)) == 0
    assert graph.value
    inst = ctypes.c_void_p()
    assert hip.hipGraphInstantiate(ctypes.byref(inst), graph, None, None, 0) == 0
    try:
        for rep in range(3):
            qs = np.stack([random_query(rng, N) for _ in range(nb)])
            q_dev.copy_(torch.from_numpy(qs.view(np.int32)))
            r1.fill_(-1), rb.fill_(-1)
            torch.cuda.synchronize()
            assert hip.hipGraphLaunch(inst, s_ptr) == 0
            stream.synchronize()
            want = [orc.row_vector_x_compressed_transposed_matrix(qs[i], dtc, N, b)[0] for i in range(nb)]
            assert np.array_equal(r1.cpu().numpy().view(np.uint32), want[0]), rep
            assert np.array_equal(rb.cpu().numpy().view(np.uint32), np.stack(want)), rep
    finally:
        hip.hipGraphExecDestroy(inst)
        hip.hipGraphDestroy(graph)
        srv.close()


def test_pack_and_respond_device_ops_replayed_from_a_graph_on_fresh_matrices(orc, device):
    """the layer-1 device ops on caller-owned memory (what replaces gpu_utils::mat_transpose + the respond the reference lacks): transpose +
    pack of D followed by a respond on the freshly packed image, captured ONCE and replayed after the host has rewritten D and q -- the pack
    pass's column sums and the response are zeroed by kernels of the library, so every replay starts clean"""
    import ctypes

    import torch

    import chalametpir_amd as cp

    rng = np.random.default_rng(78)
    N, C, b = 2 * 1536 + 5, 21, 10
    L = cp.dtc_layout_for(N, C, b, packing=2)
    stream = torch.cuda.Stream()
    D_dev = torch.zeros((N, C), dtype=torch.int32, device="cuda")
    q_dev = torch.zeros(N, dtype=torch.int32, device="cuda")
    dtc = torch.zeros(int(L.total_words) + int(L.rows_padded) + 64, dtype=torch.int32, device="cuda")
    r = torch.zeros(C, dtype=torch.int32, device="cuda")
    device.transpose_compress(D_dev, L, dtc, stream=stream)  # eager once
    device.respond(dtc, L, q_dev, r, stream=stream)
    stream.synchronize()
    hip = ctypes.CDLL("libamdhip64.so")
    s_ptr = ctypes.c_void_p(stream.cuda_stream)
    assert hip.hipStreamBeginCapture(s_ptr, 2) == 0
    try:
        device.transpose_compress(D_dev, L, dtc, stream=stream)
        device.respond(dtc, L, q_dev, r, stream=stream)
    finally:
        graph = ctypes.c_void_p()
        assert hip.hipStreamEndCapture(s_ptr, ctypes.byref(graph)) == 0
    inst = ctypes.c_void_p()
    assert graph.value and hip.hipGraphInstantiate(ctypes.byref(inst), graph, None, None, 0) == 0
    try:
        for rep in range(4):
            D = random_db_matrix(rng, N, C, b)
            q = random_query(rng, N)
            D_dev.copy_(torch.from_numpy(D.view(np.int32)))
            q_dev.copy_(torch.from_numpy(q.view(np.int32)))
            r.fill_(-1)
            torch.cuda.synchronize()
            assert hip.hipGraphLaunch(inst, s_ptr) == 0
            stream.synchronize()
            want = orc.row_vector_x_compressed_transposed_matrix(q, orc.row_wise_compress(orc.transpose(D), b), N, b)[0]
            assert np.array_equal(r.cpu().numpy().view(np.uint32), want), rep
    finally:
        hip.hipGraphExecDestroy(inst)
        hip.hipGraphDestroy(graph)
