"""The TCP demo (examples/pir_server.py = reference chalametpir_server/examples/server.rs) end to end on the GPU: a client that
speaks the reference's wire protocol (chalametpir_client/examples/client.rs:19-63) -- built here from the oracle's restatement of
chalametpir_client -- retrieves the stored value for several keys over concurrent connections."""
import asyncio
import importlib.util
import os
import socket
import struct
import threading

import numpy as np
import pytest

from _cases import unwire, wire

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_example():
    spec = importlib.util.spec_from_file_location("pir_server", os.path.join(ROOT, "examples", "pir_server.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _recv_exact(sock, n):
    buf = b""
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        assert chunk, "connection closed early"
        buf += chunk
    return buf


def test_reference_wire_protocol_end_to_end(orc, device):
    import chalametpir_amd as cp

    ex = _load_example()
    rng = np.random.default_rng(2024)
    seed = rng.bytes(32)
    server, hint, filt = cp.Server.setup(seed, ex.DEMO_DB, 3, device=device, filter_seed_material=rng.bytes(3200))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    loop = asyncio.new_event_loop()
    ready = threading.Event()

    def run():
        asyncio.set_event_loop(loop)

        async def main():
            ev = asyncio.Event()
            task = asyncio.ensure_future(ex.serve(server, seed, hint, filt, "127.0.0.1", port, ev))
            await ev.wait()
            ready.set()
            await task

        try:
            loop.run_until_complete(main())
        except asyncio.CancelledError:
            pass

    th = threading.Thread(target=run, daemon=True)
    th.start()
    assert ready.wait(30)

    results, errors = {}, []

    def client(key):
        try:
            with socket.create_connection(("127.0.0.1", port), timeout=30) as sock:
                seed_rx = _recv_exact(sock, 32)
                (n,) = struct.unpack("<I", _recv_exact(sock, 4))
                hint_rx = _recv_exact(sock, n)
                (n,) = struct.unpack("<I", _recv_exact(sock, 4))
                filt_rx = _recv_exact(sock, n)
                assert (seed_rx, hint_rx, filt_rx) == (seed, hint, filt)
                f = orc.Filter.from_bytes(filt_rx)
                H = unwire(hint_rx)
                A = orc.generate_from_seed(1774, f.num_fingerprints, seed_rx)
                crng = np.random.default_rng(abs(hash(key)) % (1 << 32))
                for _ in range(50):  # the reference retries on ArithmeticOverflowAddingQueryIndicator (test_pir.rs:66-70)
                    try:
                        qb, sc = orc.client_query(A, H, f, key, orc.ternary_vector(1774, crng), orc.ternary_vector(f.num_fingerprints, crng))
                        break
                    except orc.OracleError:
                        continue
                q = wire(qb)
                sock.sendall(struct.pack("<I", len(q)) + q)
                (n,) = struct.unpack("<I", _recv_exact(sock, 4))
                resp = unwire(_recv_exact(sock, n))
                results[key] = orc.client_process_response(f, key, sc, resp)
        except Exception as exc:  # noqa: BLE001
            errors.append((key, repr(exc)))

    keys = [b"banana", b"kiwi", b"plum", b"cantaloupe", b"apple", b"watermelon"]
    threads = [threading.Thread(target=client, args=(k,)) for k in keys]
    [t.start() for t in threads]
    [t.join(60) for t in threads]
    loop.call_soon_threadsafe(lambda: [t.cancel() for t in asyncio.all_tasks(loop)])
    th.join(10)
    assert not errors, errors
    assert results == {k: ex.DEMO_DB[k] for k in keys}
