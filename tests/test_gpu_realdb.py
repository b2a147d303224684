"""End-to-end keyword PIR on REAL encoded databases at the reference test's own ceiling (2^16 keys) and at BASELINE.json's
size (2^20 keys x (32 B, 1 kB)) -- the reference's `test_keyword_pir_with_{3,4}_wise_xor_filter`
(integrations/src/test_pir.rs:12-142) with the HIP path as the server:

    Server::setup::<ARITY>(seed, kv database)      cpir_server_setup_kv (filter construction + row encoding on the host, hint + packed
                                                   database on the GPU)
    Client::setup / query / process_response       the oracle's restatement of chalametpir_client (client.rs:39-57, 95-194, 209-275)
    Server::respond(&[u8])                         cpir_server_respond_bytes: host wire bytes in, wire bytes out

Every full-size check elsewhere in the suite is algebraic on a synthetic uniform D; here D is what the binary fuse filter really
produces (about 11 % of its rows belong to no key and are all zero at arity 3, 7 % at arity 4: matrix.rs:702-746), the queries are
LWE queries, and the assertion is the semantic one: the client gets its value back, for members, and a non-member is rejected
(client.rs:262-266).  The client's s*A is computed by the oracle for the whole batch of keys at once (or_mul streams A once per
thread); ONE key per database also goes through the oracle's literal `or_client_query` and must give the same query bytes.
"""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LWE_DIMENSION = 1774


def make_flat_db(n: int, value_bytes: int, seed: int):
    """n distinct 32-byte keys, n values of `value_bytes` bytes, as the flat arrays of cpir_kv_db"""
    rng = np.random.default_rng(seed)
    keys = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    keys[:, :8] = np.arange(n, dtype="<u8").view(np.uint8).reshape(n, 8)  # distinct
    keys[:, 8:16] ^= np.uint8(0x5A)
    values = rng.integers(0, 256, size=(n, value_bytes), dtype=np.uint8)
    key_off = np.arange(n + 1, dtype=np.uint64) * 32
    val_off = np.arange(n + 1, dtype=np.uint64) * value_bytes
    return keys, key_off, values, val_off


def wire(m: np.ndarray) -> bytes:
    m = np.ascontiguousarray(m, dtype=np.uint32)
    m = m.reshape(1, -1) if m.ndim == 1 else m
    return np.array(m.shape, dtype="<u4").tobytes() + m.tobytes()


def unwire(b: bytes) -> np.ndarray:
    rows, cols = np.frombuffer(b[:8], dtype="<u4")
    return np.frombuffer(b[8:], dtype="<u4").reshape(int(rows), int(cols))


def client_queries(orc, A, hint, filt, keys, rng):
    """Client::query for a batch of keys (client.rs:95-140 / 142-194): b = s*A + e, c = s*M, indicator added at the key's slots with the
    reference's overflow check (a key whose indicator overflows draws a fresh secret, as the reference's retry loop does,
    test_pir.rs:66-70).  Returns (queries [k x N], secrets c [k x C])."""
    N = filt.num_fingerprints
    k = len(keys)
    ind = orc.query_indicator(filt.mat_elem_bit_len)
    Q = np.empty((k, N), dtype=np.uint32)
    Cs = np.empty((k, hint.shape[1]), dtype=np.uint32)
    todo = list(range(k))
    while todo:
        S = np.stack([orc.ternary_vector(LWE_DIMENSION, rng) for _ in todo])
        B = orc.mul(S, A)  # s * A for all pending keys: parallel over the rows of S
        Cc = orc.mul(S, hint)
        again = []
        for j, i in enumerate(todo):
            q = B[j] + orc.ternary_vector_np(N, rng)  # + e (u32 wrap-around)
            ok = True
            for h in orc.filter_slots(filt, keys[i]):
                if int(q[h]) + ind >= (1 << 32):
                    ok = False  # ArithmeticOverflowAddingQueryIndicator
                    break
                q[h] += np.uint32(ind)
            if ok:
                Q[i], Cs[i] = q, Cc[j]
            else:
                again.append(i)
        todo = again
    return Q, Cs


@pytest.mark.parametrize("log_n,arity,n_queries", [(16, 3, 12), (16, 4, 12), (20, 3, 32), (20, 4, 8)])
def test_real_database_end_to_end(log_n, arity, n_queries, orc, device):
    import chalametpir_amd as cp

    t_start = time.perf_counter()
    n, value_bytes = 1 << log_n, 1024
    keys, key_off, values, val_off = make_flat_db(n, value_bytes, 0xDB0000 + 16 * log_n + arity)
    rng = np.random.default_rng(77 + log_n + arity)
    seed = rng.bytes(32)
    srv, hint_bytes, filter_bytes = cp.Server.setup_flat(seed, keys.reshape(-1), key_off, values.reshape(-1), val_off, arity, device=device)
    t_setup = time.perf_counter() - t_start
    try:
        # Client::setup (client.rs:39-57)
        filt = orc.Filter.from_bytes(filter_bytes)
        hint = unwire(hint_bytes)
        N, b = filt.num_fingerprints, filt.mat_elem_bit_len
        assert hint.shape[0] == LWE_DIMENSION and filt.arity == arity and len(filter_bytes) == 68
        assert b == cp.find_encoded_db_matrix_element_bit_length(n) and N == cp.filter_shape(arity, n)[2]
        assert srv.decompressed_num_cols == N and srv.mat_elem_bit_len == b
        if (log_n, arity) == (20, 3):  # the byte sizes the reference README publishes for this database (README.md:33-36)
            assert (len(hint_bytes), 8 + 4 * N, 8 + 4 * hint.shape[1]) == (6670248, 4718600, 3768)
        if (log_n, arity) == (20, 4):
            assert 8 + 4 * N == 4521992
        A = orc.generate_from_seed(LWE_DIMENSION, N, seed)
        # members spread over the database (first, last, random) + one key that is not in it
        idx = [0, n - 1] + [int(x) for x in rng.integers(0, n, size=n_queries - 3)]
        member_keys = [keys[i].tobytes() for i in idx]
        outsider = bytes(rng.integers(0, 256, size=32, dtype=np.uint8))
        all_keys = member_keys + [outsider]
        Q, Cs = client_queries(orc, A, hint, filt, all_keys, rng)
        # one key through the oracle's literal Client::query with the same secret material: same query, same secret
        s1, e1 = orc.ternary_vector(LWE_DIMENSION, rng), orc.ternary_vector_np(N, rng)
        try:
            q1, c1 = orc.client_query(A, hint, filt, member_keys[0], s1, e1)
            want_q = (orc.mul(s1.reshape(1, -1), A)[0] + e1).astype(np.uint32)
            for h in orc.filter_slots(filt, member_keys[0]):
                want_q[h] += np.uint32(orc.query_indicator(b))
            assert np.array_equal(q1, want_q) and np.array_equal(c1, orc.mul(s1.reshape(1, -1), hint)[0])
            resp1 = unwire(srv.respond(wire(q1)))
            assert orc.client_process_response(filt, member_keys[0], c1, resp1) == values[idx[0]].tobytes()
        except orc.OracleError as err:
            assert err.code == orc.ERR_ARITHMETIC_OVERFLOW_ADDING_QUERY_INDICATOR
        del A
        t_client = time.perf_counter() - t_start - t_setup
        # Server::respond on host wire bytes, one call per query as the reference's loop does, then the client decodes
        for j, key in enumerate(member_keys):
            resp_bytes = srv.respond(wire(Q[j]))
            assert len(resp_bytes) == 8 + 4 * hint.shape[1]
            got = orc.client_process_response(filt, key, Cs[j], unwire(resp_bytes))
            assert got == values[idx[j]].tobytes(), (j, idx[j])
        # a key the database does not hold decodes to a row without its digest (client.rs:262-266)
        with pytest.raises(orc.OracleError) as e:
            orc.client_process_response(filt, outsider, Cs[-1], unwire(srv.respond(wire(Q[-1]))))
        assert e.value.code in (orc.ERR_DECODED_ROW_NOT_PREPENDED_WITH_DIGEST, orc.ERR_ROW_NOT_DECODABLE)  # (serialization.rs:132-184 may give up first)
        # the same queries as ONE batch of concurrent callers' worth (fused passes on the device) give the same bytes
        import torch

        qd = torch.from_numpy(Q.view(np.int32)).cuda()
        rd = torch.empty((Q.shape[0], hint.shape[1]), dtype=torch.int32, device="cuda")
        srv.respond_batch_device(qd, Q.shape[0], rd, stream=torch.cuda.current_stream())
        torch.cuda.synchronize()
        rb = rd.cpu().numpy().view(np.uint32)
        for j, key in enumerate(member_keys):
            assert orc.client_process_response(filt, key, Cs[j], rb[j]) == values[idx[j]].tobytes()
        print(f"real database 2^{log_n} x {arity}-wise: setup {t_setup:.1f} s, client side {t_client:.1f} s, "
              f"{len(member_keys)} values recovered, total {time.perf_counter() - t_start:.1f} s")
    finally:
        srv.close()
