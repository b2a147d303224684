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
    shard_unit() in csrc/capi.hip -- so that neither a chunk / super-tile of the device packing, nor a packed word of the reference's
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
