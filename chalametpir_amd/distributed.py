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


def scatter_public_matrix(seed_mu: bytes, total_slots: int, unit, device=None, group=None, rows: int = 1774, block_bytes: int = 64 << 20):
    """This rank's column slab  A[:, n_g : n_{g+1}]  of the public matrix A = generate_from_seed(rows, total_slots, seed_mu)
    (reference matrix.rs:541-558, server.rs:59), with ONE expansion of the sponge per process group instead of one per rank: rank 0
    squeezes A block of rows by block of rows (the XOF is sequential: ~2 GB/s on one core) into page-locked memory and hands every other
    rank its column slab of the block while it squeezes the next one; every rank receives straight into the rows of its slab.
      * "nccl" (RCCL over xGMI): the block is uploaded once, the slabs are cut on the device and all of a block's sends go out as ONE
        batch (dist.batch_isend_irecv: one grouped launch per block instead of one per peer);
      * a host-staged backend (gloo -- the CPU tests and the one-GPU rehearsal): the slabs are cut from the page-locked block on the host
        and sent as CPU tensors; receivers upload what arrives.  (Sending device tensors through gloo copies them back to pageable host
        memory first: 162 s for 8.4 GB in a two-rank rehearsal.)
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
    slab = torch.empty((rows, hi - lo), dtype=torch.int32, device=dev)
    rb = max(1, min(rows, block_bytes // (4 * total_slots)))
    src = dist.get_global_rank(group, 0) if group is not None else 0
    peer = (lambda g: dist.get_global_rank(group, g)) if group is not None else (lambda g: g)
    depth = 4  # blocks in flight
    if rank == 0:
        xof = SeedExpander(seed_mu)
        host = [torch.empty((rb, total_slots), dtype=torch.int32) for _ in range(2)]
        if on_gpu:
            host = [h.pin_memory() for h in host]
        staged = [None, None]  # event after the upload that last read host buffer i
    elif on_gpu and via_host and hi > lo:
        landing = [torch.empty((rb, hi - lo), dtype=torch.int32).pin_memory() for _ in range(depth + 1)]
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
            xof.squeeze_into(host[b].numpy().view(np.uint32)[:n])  # releases the GIL: the previous block's transfers run meanwhile
            ops, keep = [], []
            if on_gpu and not via_host:
                blk = host[b][:n].to(dev, non_blocking=True)  # the whole block once; slabs are cut on the device
                recv.copy_(blk[:, lo:hi])
                cut = lambda a_, z_: blk[:, a_:z_].contiguous()  # noqa: E731
            else:
                if on_gpu:
                    recv.copy_(host[b][:n, lo:hi], non_blocking=True)  # rank 0's own slab, straight from the page-locked block
                else:
                    recv.copy_(host[b][:n, lo:hi])
                # (a copy, always -- .contiguous() would hand back the staging buffer itself when one rank holds every column)
                cut = lambda a_, z_: host[b][:n, a_:z_].clone(memory_format=torch.contiguous_format)  # noqa: E731
            if on_gpu:
                staged[b] = torch.cuda.Event()
                staged[b].record()
            for g in range(1, world):  # shards differ in size (ragged tail, empty shards), so point-to-point rather than dist.scatter
                a_, z_ = bounds[g]
                if z_ > a_:
                    part = cut(a_, z_)
                    keep.append(part)
                    ops.append(dist.P2POp(dist.isend, part, peer(g), group))
            works = []
            if ops:
                works = dist.batch_isend_irecv(ops) if not via_host else [dist.isend(op.tensor, op.peer, group=group) for op in ops]
            pending.append((works, keep, None))
        elif hi > lo:
            if on_gpu and via_host:
                land = landing[i % len(landing)][:n]
                pending.append(([dist.irecv(land, src=src, group=group)], land, (lambda dst=recv, s_=land: dst.copy_(s_, non_blocking=True))))
            else:
                pending.append(([dist.irecv(recv, src=src, group=group)], None, None))
        while len(pending) > depth:  # bounded queue of transfers in flight
            retire(pending.pop(0))
    while pending:
        retire(pending.pop(0))
    if on_gpu:
        torch.cuda.current_stream().synchronize()
    if rank == 0:
        xof.close()
    return slab, lo, hi


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
