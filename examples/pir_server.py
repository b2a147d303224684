#!/usr/bin/env python3
"""The reference's TCP demo server (chalametpir_server/examples/server.rs:15-95) on the MI355X path: same wire protocol, so the
reference's own client example (chalametpir_client/examples/client.rs) can talk to it unchanged.

    seed (32 B)  |  u32 LE len + hint_bytes  |  u32 LE len + filter_param_bytes   -->  client
    client  -->  u32 LE len + query_bytes
    u32 LE len + response_bytes  -->  client                       (one query per connection, as in the reference)

    python examples/pir_server.py [--port 8080] [--arity 3] [--devices 0,1,...]

`respond` runs in a thread pool: concurrent connections call it at the same time on the one handle (the reference shares an
Arc<Server> across tokio tasks); inside the library they are coalesced into batched launches.
"""
from __future__ import annotations

import argparse
import asyncio
import os
import struct
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import chalametpir_amd as cp  # noqa: E402

DEMO_DB = {  # the reference example's database (examples/server.rs:26-41)
    b"apple": b"red", b"banana": b"yellow", b"grape": b"purple", b"orange": b"orange", b"lemon": b"yellow", b"blueberry": b"blue",
    b"kiwi": b"brown", b"watermelon": b"green", b"strawberry": b"red", b"peach": b"pink", b"pineapple": b"yellow", b"cherry": b"red",
    b"avocado": b"green", b"plum": b"purple", b"cantaloupe": b"orange",
}


async def serve(server: cp.Server, seed: bytes, hint: bytes, filt: bytes, host: str, port: int, ready=None):
    loop = asyncio.get_running_loop()

    async def handle(reader: asyncio.StreamReader, writer: asyncio.StreamWriter):
        try:
            writer.write(seed + struct.pack("<I", len(hint)) + hint + struct.pack("<I", len(filt)) + filt)
            await writer.drain()
            (qlen,) = struct.unpack("<I", await reader.readexactly(4))
            query = await reader.readexactly(qlen)
            response = await loop.run_in_executor(None, server.respond, query)  # Server::respond, server.rs:184
            writer.write(struct.pack("<I", len(response)) + response)
            await writer.drain()
        except (asyncio.IncompleteReadError, ConnectionError):
            pass
        finally:
            writer.close()

    srv = await asyncio.start_server(handle, host, port)
    if ready is not None:
        ready.set()
    async with srv:
        await srv.serve_forever()


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--host", default="127.0.0.1")
    ap.add_argument("--port", type=int, default=8080)
    ap.add_argument("--arity", type=int, default=3, choices=(3, 4))
    ap.add_argument("--devices", default="0", help="comma list of device ordinals; more than one = one in-process group handle")
    args = ap.parse_args()
    seed = os.urandom(cp.SEED_BYTE_LEN)
    devs = [cp.Device(int(x)) for x in args.devices.split(",")]
    if len(devs) > 1:
        server, hint, filt = cp.Server.setup(seed, DEMO_DB, args.arity, devices=devs)
    else:
        server, hint, filt = cp.Server.setup(seed, DEMO_DB, args.arity, device=devs[0])
    print(f"PIR Server listening @ {args.host}:{args.port}", flush=True)
    asyncio.run(serve(server, seed, hint, filt, args.host, args.port))
    return 0


if __name__ == "__main__":
    sys.exit(main())
