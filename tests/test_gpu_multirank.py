"""The multi-rank path of bench.py (N-sharded database + one integer all-reduce per step) run as 2 and 3 processes on the
single GPU of the test box: every rank uses the real HIP kernels on its shard; the collective is gloo (RCCL refuses two
ranks on one device).  Rank 0 verifies the reduced responses against the oracle on the full synthetic database."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,config", [(2, "tiny"), (3, "cfg1")])
def test_bench_multirank_on_one_gpu(world, config):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", CPIR_BENCH_BACKEND="gloo", CPIR_BENCH_SHARE_DEVICE="1", OMP_NUM_THREADS="8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29600 + world), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--config", config, "--steps", "2",
           "--warmup", "1", "--no-cpu-baseline", "--verify", "--queries-per-step", "8", "--query-pool", "16"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == world and out["verified_vs_oracle"] is True and out["scaling"] == "weak"
    assert out["value"] > 0
    # the default-on proof every N > 1 line carries (no oracle involved): unit / dense / all-ones queries, both dispatch modes
    assert out["multirank_bit_exact"] is True and out["ranks_seen"] == list(range(world))
    chk = out["multirank_check"]
    assert chk["unit_queries_ok"] and chk["dense_and_all_ones_ok"] and chk["same_response_on_every_rank"] and sum(chk["shard_slots"]) > 0
    # the sharded setup (partial hints reduced to rank 0) gives the same hint as the single-process setup of the same DB
    single = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", config, "--steps", "1", "--warmup", "0",
                             "--no-cpu-baseline", "--no-live-traffic", "--queries-per-step", "8", "--query-pool", "16"], capture_output=True, text=True,
                            timeout=900)
    assert single.returncode == 0, single.stderr[-3000:]
    ref = json.loads([l for l in single.stdout.splitlines() if l.startswith("{")][-1])
    assert out["hint_checksum"] == ref["hint_checksum"] and out["server_setup_wall_sec"] > 0


def test_bench_gpus_n_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (what the driver runs for its scaling table) must start 2 ranks by
    itself -- as a child torch.distributed.run, before this process touches the GPU -- and print ONE line from rank 0"""
    env = dict(os.environ, CPIR_BENCH_BACKEND="gloo", CPIR_BENCH_SHARE_DEVICE="1", OMP_NUM_THREADS="8")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "tiny", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-setup", "--verify", "--queries-per-step", "8", "--query-pool", "16"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks"] == 2 and out["backend"] == "gloo" and out["verified_vs_oracle"] is True


def test_driver_command_rehearsal_three_ranks():
    """the driver's exact scaling command shape (`bench.py --gpus N --steps 20 --warmup 5`, its own default batch: 32 queries per GPU)
    with 3 ranks on the one GPU under the gloo hook (the pool allows 6 processes on a card: this process and the launcher hold it too, and
    one slot stays free): the line must carry its own proof of bit-exactness, all ranks seen, and the sharded setup's hint checksum"""
    env = dict(os.environ, CPIR_BENCH_BACKEND="gloo", CPIR_BENCH_SHARE_DEVICE="1", OMP_NUM_THREADS="4")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "20", "--warmup", "5", "--config", "cfg1"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    first, out = json.loads(lines[0]), json.loads(lines[-1])
    assert first["server_setup_pending"] is True and first["multirank_bit_exact"] is True  # the proof is in the line that is printed FIRST
    assert out["n_gpus"] == 3 and out["ranks"] == 3 and out["config"]["queries_per_step"] == 96 and out["steps"] == 20
    assert out["multirank_bit_exact"] is True and out["ranks_seen"] == [0, 1, 2]
    assert out["server_setup_wall_sec"] > 0 and "server_setup_timed_out" not in out and out["hint_checksum"] > 0
    # like for like: the shards of this Infinity-Cache-sized database run their passes interleaved; the line also carries the same steps in
    # slice order (already in the line printed first) and rank 0's single-GPU run of the whole database, and divides the two
    assert out["roofline"]["pass_order"] == "interleaved" and first["value_slice_order"] > 0 and first["slice_order"]["frac"] > 0
    ref = out["single_gpu_reference"]
    assert ref["queries_per_sec"] > 0 and ref["pass_order"] == "slice" and ref["queries_per_step"] == 32
    assert abs(out["scaling_like_for_like"] - out["value_slice_order"] / ref["queries_per_sec"]) < 2e-3
    assert abs(out["scaling_as_dispatched"] - out["value"] / ref["queries_per_sec"]) < 2e-3


def test_setup_deadline_is_not_a_success():
    """a sharded setup whose collectives never come back: the respond line is out first, the LAST line says so, the exit code is not 0"""
    env = dict(os.environ, CPIR_BENCH_BACKEND="gloo", CPIR_BENCH_SHARE_DEVICE="1", OMP_NUM_THREADS="4", CPIR_BENCH_TEST_HANG_SETUP="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--config", "tiny", "--setup-deadline", "5"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0, p.stdout[-2000:]
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert lines[0]["server_setup_pending"] is True and lines[0]["value"] > 0
    assert lines[-1]["server_setup_timed_out"] is True and "deadline" in lines[-1]["server_setup_error"] and lines[-1]["value"] == lines[0]["value"]


def test_bench_single_rank_verify():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "cfg1", "--steps", "2", "--warmup", "1", "--no-setup",
           "--no-cpu-baseline", "--no-live-traffic", "--verify", "--group-shards", "3"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert out["verified_vs_oracle"] is True and out["roofline"]["bound"] == "hbm"
    # the in-process group path of the bench (three shards on the one GPU): same responses as the single-device server
    grp = out["respond_host_path_group"]
    assert "error" not in grp and len(grp["shards"]) == 3 and grp["responses_equal_single_device"] is True
    assert out["batched_respond"]["queries_per_sec"] > 0 and out["respond_host_path"]["one_caller_queries_per_sec"] > 0


def test_bench_json_contract():
    """the one JSON line the driver parses: every key of the bench contract, with the types and relations it relies on"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "cfg1", "--steps", "3", "--warmup", "1", "--no-setup",
           "--no-host-path", "--cpu-seconds", "1"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "rank 0 prints ONE line on stdout"
    d = json.loads(lines[0])
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict)):
        assert isinstance(d[key], typ), key
    assert d["vs_baseline"] is None and d["higher_is_better"] is True and d["scaling"] in ("weak", "strong")
    assert (d["n_gpus"], d["steps"], d["warmup"], d["data"], d["dtype"]) == (1, 3, 1, "synthetic", "u32")
    assert "workload" in d["config"] and "model" not in d["config"]
    roof = d["roofline"]
    assert roof["bound"] in ("hbm", "mfma") and roof["unit"] in ("GB/s", "TFLOP/s") and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3 and "traffic" in roof
    # HBM traffic per launch is MEASURED in the run (two child runs of the timed loop under rocprofv3 --pmc): this Infinity-Cache-sized
    # database is walked in the interleaved order (its 32 passes share one stream of it on die), so anything between 1/32 of the layout
    # bytes plus the queries and a little above all of them
    assert isinstance(roof["traffic"], int) and roof["traffic"] > 0 and "measured in this run" in roof["traffic_source"]
    assert 0.01 < roof["traffic_over_moved_bytes"] < 1.2 and roof["pass_order"] == "interleaved"
    # the bytes really moved (the resident layout is tighter than the reference packing), against spec and against a live read-only probe
    assert abs(roof["frac_moved"] - roof["moved_GBps"] / roof["peak"]) < 1e-3 and roof["frac_moved"] <= roof["frac"] + 1e-3
    assert roof["read_ceiling_GBps"] > 1000 and abs(roof["frac_vs_read_ceiling"] - roof["moved_GBps"] / roof["read_ceiling_GBps"]) < 1e-3
    assert d["ranks"] == 1 and d["backend"] is None
    assert abs(d["value"] - d["config"]["queries_per_step"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.01
    cpu = d["cpu_baseline"]
    assert cpu["kind"] in ("reference", "port") and cpu["cores"] >= 1 and cpu["value"] > 0 and isinstance(cpu["sample"], str)
    assert cpu["gpu_results_bit_exact"] is True


def test_bench_default_sections_at_the_reference_tests_ceiling():
    """the sections the driver's default run carries besides the headline, at 2^16 keys so that it takes seconds: Server::setup from the
    key-value database (the reference's own `server_setup` bench), the real database it built served without its empty rows and looked
    up end to end, the pack pass against a plain mixed stream"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "cfg1", "--steps", "4", "--warmup", "1", "--no-cpu-baseline",
           "--no-live-traffic", "--no-host-path"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert d["server_setup_kv_wall_sec"] > 0 and d["server_setup_kv_phases_sec"]["encode"] > 0 and d["hint_bytes"] == 8 + 4 * 1774 * 846
    real = d["real_db"]
    assert real["rows_owned_by_no_key"] == 77824 - 65536 and real["slots_served"] == 65536 and real["slots_of"] == 77824
    assert real["queries_per_sec"] > 0 and real["end_to_end"]["all_recovered"] is True and real["end_to_end"]["keys_looked_up"] >= 1
    tc = d["setup_roofline"]["transpose_compress"]
    assert "planar_pack" in tc["kernel"] and tc["copy_ceiling_GBps"] > 1000 and 0.05 < tc["frac_vs_copy_ceiling"] < 1.3


def test_rccl_backend_single_rank():
    """the backend bench.py --gpus N really uses ("nccl" = RCCL), as far as a one-GPU box can take it: one rank, device tensors, the
    product's own collectives and the device path of scatter_public_matrix (tests/_rccl_worker.py)"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", "29655", os.path.join(ROOT, "tests", "_rccl_worker.py")]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "rccl single rank ok" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


def test_two_ranks_over_the_real_backend_where_the_box_has_two_devices():
    """ARMS ITSELF on a box with two or more GPUs (the driver's 8-GPU node): `bench.py --gpus 2` over the real backend -- "nccl" = RCCL,
    one rank per device -- must bring up its process group, reduce the shards' partial responses with the int32-view all-reduce and prove
    the result bit-exact (`multirank_bit_exact`), and the sharded setup must give the single-process hint.  On a one-GPU box the same
    command runs under the gloo / shared-device hook (RCCL refuses two ranks on one device): never skipped."""
    import torch

    two = torch.cuda.device_count() >= 2
    env = dict(os.environ, OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if not two:
        env.update(CPIR_BENCH_BACKEND="gloo", CPIR_BENCH_SHARE_DEVICE="1")
    else:
        env.pop("CPIR_BENCH_BACKEND", None)
        env.pop("CPIR_BENCH_SHARE_DEVICE", None)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--config", "cfg1", "--verify"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["ranks"] == 2 and out["backend"] == ("nccl" if two else "gloo")
    assert out["multirank_bit_exact"] is True and out["ranks_seen"] == [0, 1] and out["verified_vs_oracle"] is True
    assert out["server_setup_wall_sec"] > 0 and out["hint_checksum"] > 0 and out["scaling_like_for_like"] > 0
