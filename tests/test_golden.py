"""Committed golden fixtures (tests/golden/*.npz, minted by tests/golden/make_golden.py):
  * CPU: the oracle still reproduces them (it cannot drift silently);
  * GPU: the HIP path reproduces them through the C ABI without needing anything but the fixture data."""
import os

import numpy as np
import pytest

from _cases import cf_of, unwire, wire

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name))


def respond_keys(z):
    return sorted({k.rsplit("_", 1)[0] for k in z.files})


def test_oracle_reproduces_respond_fixtures(orc):
    z = load("respond_cases.npz")
    keys = respond_keys(z)
    assert len(keys) == 2 * 4 + 3 * 2 + 4 * 5  # every b in 4..14 x every N mod cf
    for k in keys:
        b = int(k.split("_")[0][1:])
        D, q, dtc, r = z[k + "_D"], z[k + "_q"], z[k + "_dtc"], z[k + "_r"]
        assert np.array_equal(orc.row_wise_compress(orc.transpose(D), b), dtc)
        assert np.array_equal(orc.row_vector_x_compressed_transposed_matrix(q, dtc, D.shape[0], b)[0], r)


def test_oracle_reproduces_setup_fixtures(orc):
    z = load("setup_cases.npz")
    for i in range(4):
        k = f"s{i}"
        b, seed = int(z[k + "_b"][0]), z[k + "_seed"].tobytes()
        hint, dtc = orc.server_setup_from_matrix(seed, z[k + "_D"], b)
        assert np.array_equal(hint, z[k + "_hint"]) and np.array_equal(dtc, z[k + "_dtc"])
        assert orc.server_respond(dtc, z[k + "_D"].shape[0], b, wire(z[k + "_q"])) == z[k + "_resp"].tobytes()


def _kv(z, k):
    def split(buf, lens):
        out, o = [], 0
        for n in lens:
            out.append(buf[o:o + int(n)].tobytes())
            o += int(n)
        return out

    return split(z[k + "_keys"], z[k + "_klen"]), split(z[k + "_vals"], z[k + "_vlen"])


def test_oracle_and_product_encoder_reproduce_kv_fixtures(orc, native):
    """the product's HOST encoder needs no GPU, so it is checked against the fixtures here as well"""
    import chalametpir_amd as cp

    z = load("kv_cases.npz")
    for arity in (3, 4):
        k = f"kv{arity}"
        keys, vals = _kv(z, k)
        b, fseeds = int(z[k + "_b"][0]), z[k + "_fseeds"].tobytes()
        D, filt, _ = orc.from_kv_database(arity, keys, vals, b, fseeds)
        assert np.array_equal(D, z[k + "_D"]) and filt.to_bytes() == z[k + "_filter"].tobytes()
        D2, fbytes = cp.encode_kv_database(dict(zip(keys, vals)), arity, b, fseeds)
        assert np.array_equal(D2, z[k + "_D"]) and fbytes == z[k + "_filter"].tobytes()


@pytest.mark.gpu
def test_gpu_reproduces_respond_fixtures(device):
    import chalametpir_amd as cp

    z = load("respond_cases.npz")
    for k in respond_keys(z):
        b = int(k.split("_")[0][1:])
        D, q, dtc, r = z[k + "_D"], z[k + "_q"], z[k + "_dtc"], z[k + "_r"]
        srv = cp.Server.from_compressed(dtc, D.shape[0], b, device=device)
        assert np.array_equal(srv.respond_array(q), r), k
        assert np.array_equal(srv.export_compressed(), dtc), k


@pytest.mark.gpu
def test_gpu_reproduces_setup_fixtures(device):
    import chalametpir_amd as cp

    z = load("setup_cases.npz")
    for i in range(4):
        k = f"s{i}"
        b, seed = int(z[k + "_b"][0]), z[k + "_seed"].tobytes()
        srv, hint = cp.Server.setup_from_matrix(seed, z[k + "_D"], b, device=device)
        assert np.array_equal(hint, z[k + "_hint"]), k
        assert np.array_equal(srv.export_compressed(), z[k + "_dtc"]), k
        assert srv.respond(wire(z[k + "_q"])) == z[k + "_resp"].tobytes(), k


@pytest.mark.gpu
def test_gpu_reproduces_kv_fixtures(device):
    import chalametpir_amd as cp

    z = load("kv_cases.npz")
    for arity in (3, 4):
        k = f"kv{arity}"
        keys, vals = _kv(z, k)
        srv, hint_bytes, fbytes = cp.Server.setup(bytes(range(32)), dict(zip(keys, vals)), arity, device=device,
                                                  filter_seed_material=z[k + "_fseeds"].tobytes())
        assert fbytes == z[k + "_filter"].tobytes()
        b = int(z[k + "_b"][0])
        cf = cf_of(b)
        D = z[k + "_D"]
        # packed DB = compress(transpose(D)) computed here with numpy only (no oracle): field j of word w is D[cf*w+j]
        N, C = D.shape
        W = -(-N // cf)
        pad = np.zeros((W * cf, C), dtype=np.uint32)
        pad[:N] = D & ((1 << b) - 1)
        want = np.zeros((C, W), dtype=np.uint32)
        for j in range(cf):
            want |= (pad[j::cf].T << np.uint32(j * (32 // cf))).astype(np.uint32)
        assert np.array_equal(srv.export_compressed(), want)
        assert unwire(hint_bytes).shape == (1774, C)
