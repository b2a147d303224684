"""Worker for tests/test_gpu_multirank.py::test_rccl_backend_single_rank: ONE rank under the "nccl" backend (= RCCL on ROCm) on the box's
one GPU.  Two ranks cannot share a device under RCCL, so this is what a one-GPU box can say about the backend bench.py --gpus N uses:
the process group comes up (dmabuf IPC mode), the int32-view sum collectives the product issues run on device tensors, and
scatter_public_matrix takes its device path (block uploaded once, slab cut on the device)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import chalametpir_amd as cp  # noqa: E402
from chalametpir_amd.distributed import ShardedServer, allreduce_u32_, reduce_u32_, scatter_public_matrix, shard_range  # noqa: E402


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    dev = cp.Device(0)
    stream = torch.cuda.current_stream()
    N, C, b = 3 * 1536 + 5, 21, 9
    D = torch.empty((N, C), dtype=torch.int32, device="cuda")
    dev.synth_fill(D, N * C, 7, mask=(1 << b) - 1, stream=stream)
    lo, hi = shard_range(N, cp.dtc_layout_for(N, C, b), 0, 1)
    srv = ShardedServer.from_device_matrix(D, lo, hi, C, b, N, dev, stream=stream)
    q = torch.empty(N, dtype=torch.int32, device="cuda")
    dev.synth_fill(q, N, 9, stream=stream)
    r = torch.empty(C, dtype=torch.int32, device="cuda")
    work = srv.respond_device(q, r, stream=stream, async_op=True)  # partial mat-vec + the all-reduce bench.py overlaps with the next step
    work.wait()
    torch.cuda.synchronize()
    want = ((q.to(torch.int64) & 0xFFFFFFFF)[:, None] * D.to(torch.int64)).sum(dim=0) & 0xFFFFFFFF
    assert torch.equal(r.to(torch.int64) & 0xFFFFFFFF, want)
    t = torch.arange(-5, 5, dtype=torch.int32, device="cuda")
    keep = t.clone()
    allreduce_u32_(t)
    reduce_u32_(t, dst=0)
    torch.cuda.synchronize()
    assert torch.equal(t, keep)
    seed = bytes(range(32))
    slab, a, z = scatter_public_matrix(seed, 2000, 512, device=torch.device("cuda", 0), rows=37, block_bytes=4 * 2000 * 5)
    assert (a, z) == (0, 2000)
    assert np.array_equal(slab.cpu().numpy().view(np.uint32), cp.generate_from_seed(37, 2000, seed))
    # every other collective bench.py --gpus N issues, with the dtypes and reduce ops it uses (one rank: the values must come back unchanged,
    # but the backend has to accept the dtype / op combination): wall-time MAX in float64, int64 SUM / MIN / MAX of the multirank check,
    # the step's asynchronous int32 all-reduce with its wait
    t64 = torch.tensor([1.25, 2.5], dtype=torch.float64, device="cuda")
    dist.all_reduce(t64, op=dist.ReduceOp.MAX)
    i64 = torch.tensor([3, -4, 1 << 40], dtype=torch.int64, device="cuda")
    for op in (dist.ReduceOp.SUM, dist.ReduceOp.MIN, dist.ReduceOp.MAX):
        v = i64.clone()
        dist.all_reduce(v, op=op)
        assert torch.equal(v, i64), op
    big = torch.arange(256 * 940, dtype=torch.int32, device="cuda").reshape(256, 940)
    work = dist.all_reduce(big, async_op=True)
    work.wait()
    torch.cuda.synchronize()
    assert t64.tolist() == [1.25, 2.5] and int(big[255, 939]) == 256 * 940 - 1
    dist.barrier()
    dist.destroy_process_group()
    print("rccl single rank ok")


if __name__ == "__main__":
    main()
