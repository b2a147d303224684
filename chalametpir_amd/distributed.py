"""Sharding the database across the GPUs of one node: one process per GPU, `torch.distributed` (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" in the CPU tests).

The reference has no multi-device code at all.  Both contractions of the hot path sum over the filter
slots n in [0, N), so the database is partitioned along n ("row-partitioning the encoded DB D" = column slabs of the
stored D^T): rank g holds packed words for slots [n_g, n_{g+1}) and computes a full-length PARTIAL response / hint from
its slice of the query / of A.  The one exchange step is an integer SUM over ranks; u32 wrap-around addition is
associative and commutative, so the reduced result is bit-identical to the single-GPU result whatever the reduction order.

  respond : r   = sum_g  q[n_g:n_{g+1}] . D[n_g:n_{g+1}, :]         all_reduce / reduce of C  u32   (3.7 KB at 1 kB values)
  setup   : M   = sum_g  A[:, n_g:n_{g+1}] . D[n_g:n_{g+1}, :]      reduce of 1774 x C u32          (6.7 MB)

RCCL has no unsigned-32 sum on every build, but two's-complement int32 addition produces the same bits, so tensors are
reduced through an int32 view.
"""
from __future__ import annotations

from typing import Optional, Tuple

from .server import Device, Server

def shard_unit(layout) -> int:
    """Granularity of shard boundaries for a `cpir_dtc_layout`: lcm(slots_per_chunk, compression_factor) -- the same rule as
    shard_unit() in csrc/host_setup.hip -- so that neither a chunk / super-tile of the device packing, nor a packed word of the reference's
    representation (import / export), nor a 16-byte query load straddles two shards."""
    import ctypes

    from . import _native

    return int(_native.load().cpir_shard_unit(ctypes.byref(layout)))


def shard_range(total_slots: int, unit, rank: int, world_size: int) -> Tuple[int, int]:
    """Slots [begin, end) held by `rank`.  `unit` is a `cpir_dtc_layout` (boundaries are then multiples of shard_unit(layout)) or
    that number itself; the last shard takes the ragged tail.  Shards may be empty when there are fewer units than ranks."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    unit = shard_unit(unit) if hasattr(unit, "slots_per_chunk") else int(unit)
    n_units = -(-total_slots // unit)
    lo = (n_units * rank // world_size) * unit
    hi = (n_units * (rank + 1) // world_size) * unit
    return min(lo, total_slots), min(hi, total_slots)


def allreduce_u32_(t, group=None, async_op: bool = False):
    """in-place wrap-around sum of a 4-byte-element tensor over the process group"""
    import torch
    import torch.distributed as dist

    view = t if t.dtype == torch.int32 else t.view(torch.int32)
    return dist.all_reduce(view, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


def reduce_u32_(t, dst: int = 0, group=None):
    import torch
    import torch.distributed as dist

    view = t if t.dtype == torch.int32 else t.view(torch.int32)
    return dist.reduce(view, dst=dst, op=dist.ReduceOp.SUM, group=group)


def scatter_public_matrix(seed_mu: bytes, total_slots: int, unit, device=None, group=None, rows: int = 1774, block_bytes: int = 64 << 20,
                          via: Optional[str] = None):
    """This rank's column slab  A[:, n_g : n_{g+1}]  of the public matrix A = generate_from_seed(rows, total_slots, seed_mu)
    (reference matrix.rs:541-558, server.rs:59), with ONE expansion of the sponge per process group instead of one per rank: rank 0
    squeezes A block of rows by block of rows (the XOF is sequential: ~2 GB/s on one core) into page-locked memory and hands every other
    rank its columns of the block while it squeezes the next one.  Two transports, same result:
      * via="broadcast" (the default under "nccl" = RCCL over xGMI): the block is uploaded once and BROADCAST whole -- a plain collective on
        the group's own communicator, stream-ordered, nothing point-to-point (round 3 sent slabs with batch_isend_irecv against lone irecvs:
        two different communicators on a lazily initialised group, i.e. a deadlock waiting for a caller) -- and every rank cuts its slab
        out of the landed block on its device.  8x the bytes of a scatter on the links, and still nothing: 8.4 GB per setup at 2^20 keys
        against a sponge that takes 4.2 s to produce them;
      * via="p2p" (the default under a host-staged backend, gloo: the CPU tests and the one-GPU rehearsal): the slabs are cut from the
        page-locked block on the host and sent point-to-point as CPU tensors, isend against irecv; receivers upload what arrives.
        (Sending device tensors through gloo copies them back to pageable host memory first: 162 s for 8.4 GB in a two-rank rehearsal.)
    `unit` as in shard_range.  Returns (slab [rows x n_g] int32 on `device`, lo, hi).  device=None: CPU tensors (the CPU test)."""
    import numpy as np
    import torch
    import torch.distributed as dist

    from .server import SeedExpander

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    bounds = [shard_range(total_slots, unit, r, world) for r in range(world)]
    lo, hi = bounds[rank]
    dev = torch.device("cpu") if device is None else torch.device(device)
    on_gpu = dev.type == "cuda"
    via_host = dist.get_backend(group) != "nccl"  # the collective moves host memory
    if via is None:
        via = "p2p" if via_host else "broadcast"
    if via not in ("p2p", "broadcast") or (via == "p2p" and not via_host):
        raise ValueError("via must be 'broadcast' (any backend) or 'p2p' (host-staged backends only)")
    slab = torch.empty((rows, hi - lo), dtype=torch.int32, device=dev)
    rb = max(1, min(rows, block_bytes // (4 * total_slots)))
    src = dist.get_global_rank(group, 0) if group is not None else 0
    peer = (lambda g: dist.get_global_rank(group, g)) if group is not None else (lambda g: g)
    xof = SeedExpander(seed_mu) if rank == 0 else None
    try:
        if via == "broadcast":
            _scatter_by_broadcast(xof, slab, lo, hi, rows, rb, total_slots, rank, src, group, dev, on_gpu, via_host)
        else:
            _scatter_by_p2p(xof, slab, bounds, rows, rb, total_slots, rank, world, src, peer, group, dev, on_gpu)
    finally:
        if xof is not None:
            xof.close()
    if on_gpu:
        torch.cuda.current_stream().synchronize()
    return slab, lo, hi


def _scatter_by_broadcast(xof, slab, lo, hi, rows, rb, total_slots, rank, src, group, dev, on_gpu, via_host):
    """whole blocks of rows broadcast from rank 0; every rank keeps its columns.  With a device backend everything is stream-ordered
    (a non-async collective is enqueued behind the current stream and the current stream waits for it): the only host-side wait is rank
    0's, before it overwrites a page-locked block whose upload may still be reading it."""
    import numpy as np
    import torch
    import torch.distributed as dist

    staged_on_host = (not on_gpu) or via_host  # the collective carries CPU tensors
    n_buf = 2
    if rank == 0:
        host = [torch.empty((rb, total_slots), dtype=torch.int32) for _ in range(n_buf)]
        if on_gpu:
            host = [h.pin_memory() for h in host]
        uploaded = [None] * n_buf
    if not staged_on_host:
        land = [torch.empty((rb, total_slots), dtype=torch.int32, device=dev) for _ in range(n_buf)]
    elif rank != 0:
        land = [torch.empty((rb, total_slots), dtype=torch.int32) for _ in range(n_buf)]
        if on_gpu:
            land = [h.pin_memory() for h in land]
        copied = [None] * n_buf
    for i, r0 in enumerate(range(0, rows, rb)):
        n = min(rb, rows - r0)
        b = i % n_buf
        recv = slab[r0:r0 + n]
        if rank == 0:
            if uploaded[b] is not None:
                uploaded[b].synchronize()
                uploaded[b] = None
            xof.squeeze_into(host[b].numpy().view(np.uint32)[:n])  # releases the GIL
            if staged_on_host:
                if on_gpu:
                    recv.copy_(host[b][:n, lo:hi], non_blocking=True)
                    uploaded[b] = torch.cuda.Event()
                    uploaded[b].record()
                else:
                    recv.copy_(host[b][:n, lo:hi])
                dist.broadcast(host[b][:n], src=src, group=group)
            else:
                land[b][:n].copy_(host[b][:n], non_blocking=True)
                uploaded[b] = torch.cuda.Event()
                uploaded[b].record()
                dist.broadcast(land[b][:n], src=src, group=group)
                recv.copy_(land[b][:n, lo:hi])
        else:
            if staged_on_host and on_gpu and copied[b] is not None:  # the upload that last read this landing block
                copied[b].synchronize()
                copied[b] = None
            dist.broadcast(land[b][:n], src=src, group=group)
            if hi > lo:
                if staged_on_host and on_gpu:
                    recv.copy_(land[b][:n, lo:hi], non_blocking=True)
                    copied[b] = torch.cuda.Event()
                    copied[b].record()
                else:
                    recv.copy_(land[b][:n, lo:hi])
    if on_gpu:
        torch.cuda.current_stream().synchronize()


def _scatter_by_p2p(xof, slab, bounds, rows, rb, total_slots, rank, world, src, peer, group, dev, on_gpu):
    """host-staged backends: slabs cut from rank 0's page-locked block, isend against irecv (both lone, the same communicator)"""
    import numpy as np
    import torch
    import torch.distributed as dist

    lo, hi = bounds[rank]
    depth = 4  # blocks in flight
    if rank == 0:
        host = [torch.empty((rb, total_slots), dtype=torch.int32) for _ in range(2)]
        if on_gpu:
            host = [h.pin_memory() for h in host]
        staged = [None, None]  # event after the upload that last read host buffer i
    elif on_gpu and hi > lo:
        landing = [torch.empty((rb, hi - lo), dtype=torch.int32).pin_memory() for _ in range(depth + 1)]
        landed = [None] * (depth + 1)  # event behind the upload that last read landing block i
    pending = []  # (works, what they must outlive, upload to run once they are done)

    def retire(entry):
        works, _keep, then = entry
        for w in works:
            w.wait()
        if then is not None:
            then()

    for i, r0 in enumerate(range(0, rows, rb)):
        n = min(rb, rows - r0)
        recv = slab[r0:r0 + n]  # n whole rows of the slab: contiguous
        if rank == 0:
            b = i & 1
            if staged[b] is not None:  # the upload that read this host buffer two blocks ago must be done before it is overwritten
                staged[b].synchronize()
                staged[b] = None
            # (sends of this buffer from two blocks ago were of CLONES, so only the upload above reads the buffer itself)
            xof.squeeze_into(host[b].numpy().view(np.uint32)[:n])  # releases the GIL: the previous block's transfers run meanwhile
            if on_gpu:
                recv.copy_(host[b][:n, lo:hi], non_blocking=True)  # rank 0's own slab, straight from the page-locked block
                staged[b] = torch.cuda.Event()
                staged[b].record()
            else:
                recv.copy_(host[b][:n, lo:hi])
            works, keep = [], []
            for g in range(1, world):  # shards differ in size (ragged tail, empty shards), so point-to-point rather than dist.scatter
                a_, z_ = bounds[g]
                if z_ > a_:
                    # (a copy, always -- .contiguous() would hand back the staging buffer itself when one rank holds every column)
                    part = host[b][:n, a_:z_].clone(memory_format=torch.contiguous_format)
                    keep.append(part)
                    works.append(dist.isend(part, peer(g), group=group))
            pending.append((works, keep, None))
        elif hi > lo:
            if on_gpu:
                slot = i % len(landing)
                if landed[slot] is not None:  # its upload must be through before the next receive writes into the block
                    landed[slot].synchronize()
                    landed[slot] = None
                land = landing[slot][:n]

                def upload(dst=recv, s_=land, slot=slot):
                    dst.copy_(s_, non_blocking=True)
                    landed[slot] = torch.cuda.Event()
                    landed[slot].record()

                pending.append(([dist.irecv(land, src=src, group=group)], land, upload))
            else:
                pending.append(([dist.irecv(recv, src=src, group=group)], None, None))
        while len(pending) > depth:  # bounded queue of transfers in flight
            retire(pending.pop(0))
    while pending:
        retire(pending.pop(0))


class ShardedServer:
    """One rank's shard of a database plus the collective that completes a response.

    `respond_device(q, out)` enqueues the local partial mat-vec on the current stream and sum-reduces `out` (C x int32)
    across ranks; with `batch > 1` the partials of a whole batch of in-flight queries are reduced by ONE collective
    (the message is tiny -- 4*C bytes per query -- so the reduce is latency-bound and batching amortises it)."""

    def __init__(self, local: Optional[Server], num_cols: int, total_slots: int, group=None):
        self.local = local  # None when this rank's shard is empty
        self.num_cols = num_cols
        self.total_slots = total_slots
        self.group = group

    @staticmethod
    def from_device_matrix(D_shard, slot_begin: int, slot_end: int, num_cols: int, mat_elem_bit_len: int, total_slots: int,
                           device: Device, group=None, stream=None) -> "ShardedServer":
        local = None
        if slot_end > slot_begin:
            local = Server.from_device_matrix(D_shard, slot_end - slot_begin, num_cols, mat_elem_bit_len, device=device,
                                              slot_offset=slot_begin, total_slots=total_slots, stream=stream)
        return ShardedServer(local, num_cols, total_slots, group)

    def respond_partial_device(self, q_dev, out_dev, batch: int = 1, stream=None) -> None:
        if self.local is None:
            out_dev.zero_()
        elif batch == 1:
            self.local.respond_device(q_dev, out_dev, stream=stream)
        else:
            self.local.respond_batch_device(q_dev, batch, out_dev, stream=stream)

    def respond_device(self, q_dev, out_dev, batch: int = 1, stream=None, async_op: bool = False):
        self.respond_partial_device(q_dev, out_dev, batch=batch, stream=stream)
        return allreduce_u32_(out_dev, group=self.group, async_op=async_op)
