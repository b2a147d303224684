"""Worker for tests/test_distributed_cpu.py: one rank of a world_size-N gloo job (launched by torch.distributed.run).

No GPU here, so each rank's PARTIAL response / hint over its shard comes from the CPU oracle; what is under test is the
product's sharding + exchange logic (chalametpir_amd.distributed: shard_range, allreduce_u32_, reduce_u32_), i.e. that
the N-partition plus an int32-view sum reproduces the unsharded u32 wrap-around result bit for bit."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from chalametpir_amd.distributed import allreduce_u32_, reduce_u32_, scatter_public_matrix, shard_range  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from _cases import cf_of, random_db_matrix, random_query  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    for case, (b, N, C) in enumerate(((9, 3 * 1024 * 5 + 7, 33), (12, 2 * 1024 * 3, 8), (6, 4 * 1024 + 1, 5), (10, 100, 4))):
        rng = np.random.default_rng(1234 + case)  # same data on every rank
        cf = cf_of(b)
        D = random_db_matrix(rng, N, C, b)
        q = random_query(rng, N)
        A = random_query(rng, 16 * N).reshape(16, N)  # a 16-row slice of "A" is enough to exercise the hint reduce
        want_r = orc.row_vector_x_compressed_transposed_matrix(q, orc.row_wise_compress(orc.transpose(D), b), N, b)[0]
        want_m = orc.mul(A, D)

        lo, hi = shard_range(N, cf * 1024, rank, world)
        # shards tile [0, N) without gaps or overlaps and start on packing-unit boundaries
        bounds = [shard_range(N, cf * 1024, r, world) for r in range(world)]
        assert bounds[0][0] == 0 and bounds[-1][1] == N
        assert all(bounds[i][1] == bounds[i + 1][0] for i in range(world - 1))
        assert all(lo_ % (cf * 1024) == 0 for lo_, _ in bounds)

        if hi > lo:
            dtc_shard = orc.row_wise_compress(orc.transpose(D[lo:hi]), b)
            part_r = orc.row_vector_x_compressed_transposed_matrix(q[lo:hi], dtc_shard, hi - lo, b)[0]
            part_m = orc.mul(A[:, lo:hi], D[lo:hi])
        else:
            part_r = np.zeros(C, dtype=np.uint32)
            part_m = np.zeros((16, C), dtype=np.uint32)

        t = torch.from_numpy(part_r.view(np.int32).copy())
        allreduce_u32_(t)
        assert np.array_equal(t.numpy().view(np.uint32), want_r), (case, rank)

        m = torch.from_numpy(part_m.view(np.int32).copy())
        reduce_u32_(m, dst=0)
        if rank == 0:
            assert np.array_equal(m.numpy().view(np.uint32), want_m), case
    # ONE expansion of A per group (rank 0 squeezes, column slabs scattered): every rank ends up with exactly its columns of
    # generate_from_seed(rows, N, seed), and the partial hints over those slabs sum to A * D
    seed = bytes(range(32))
    for rows, N, C, unit, block_bytes in ((50, 3 * 1024 * 4 + 5, 6, 3 * 1024, 4 * (3 * 1024 * 4 + 5) * 7), (1774, 2000, 3, 512, 64 << 20),
                                          (9, 700, 2, 1024, 1)):
        for via in ("p2p", "broadcast", None):  # both transports (None = the backend's default: p2p under gloo)
            slab, lo, hi = scatter_public_matrix(seed, N, unit, rows=rows, block_bytes=block_bytes, via=via)
            assert (lo, hi) == shard_range(N, unit, rank, world)
            A = orc.generate_from_seed(rows, N, seed)
            assert np.array_equal(slab.numpy().view(np.uint32), A[:, lo:hi]), (rows, N, rank)
            D = random_db_matrix(np.random.default_rng(7), N, C, 9)
            part = orc.mul(np.ascontiguousarray(A[:, lo:hi]), D[lo:hi]) if hi > lo else np.zeros((rows, C), dtype=np.uint32)
            m = torch.from_numpy(part.view(np.int32).copy())
            reduce_u32_(m, dst=0)
            if rank == 0:
                assert np.array_equal(m.numpy().view(np.uint32), orc.mul(A, D)), (rows, N)
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} ok")


if __name__ == "__main__":
    main()
