"""world_size-2, -3 and -8 gloo runs of the N-sharding + integer-sum exchange (the multi-GPU path of SURVEY.md 8e), on CPU."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_shard_range_properties():
    from chalametpir_amd.distributed import shard_range

    for N in (1, 100, 3071, 3072, 3073, 1_179_648, 4_718_592):
        for cf in (2, 3, 4):
            for world in (1, 2, 3, 4, 8):
                b = [shard_range(N, cf * 1024, r, world) for r in range(world)]
                assert b[0][0] == 0 and b[-1][1] == N
                for i in range(world):
                    lo, hi = b[i]
                    assert 0 <= lo <= hi <= N and lo % (cf * 1024) == 0
                    if i:
                        assert lo == b[i - 1][1]
    # the bench config splits evenly over 8 GPUs
    assert [hi - lo for lo, hi in (shard_range(1_179_648, 3 * 1024, r, 8) for r in range(8))] == [147_456] * 8
    # dense64 at b = 9: 7 * 1024 slots per chunk, 165 chunks (the last one ragged) over 8 ranks
    sizes = [hi - lo for lo, hi in (shard_range(1_179_648, 7 * 1024, r, 8) for r in range(8))]
    assert sum(sizes) == 1_179_648 and max(sizes) - min(sizes) <= 7 * 1024 + 4096
    with pytest.raises(ValueError):
        shard_range(10, 3 * 1024, 2, 2)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_exchange_under_gloo(world, orc):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    port = 29511 + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(HERE, "_dist_worker.py")]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert p.stdout.count("ok") == world
