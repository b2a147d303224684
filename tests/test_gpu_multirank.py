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
    itself -- as a child torch.distributed.run, before this process touches the GPU -- and only rank 0 prints"""
    env = dict(os.environ, CPIR_BENCH_BACKEND="gloo", CPIR_BENCH_SHARE_DEVICE="1", OMP_NUM_THREADS="8")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "tiny", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-setup", "--verify", "--queries-per-step", "8", "--query-pool", "16"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    # (rank 0 prints the line after the timed region and again after every stage behind it: a consumer takes the LAST one, which has
    # nothing pending)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]  # (the gloo test backend's own C++ chatter also lands on stdout)
    out = json.loads(lines[-1])
    assert not any(k.endswith("_pending") for k in out) and json.loads(lines[0])["value"] == out["value"]
    assert out["n_gpus"] == 2 and out["ranks"] == 2 and out["backend"] == "gloo" and out["verified_vs_oracle"] is True
    assert out["single_gpu_reference"]["queries_per_sec"] > 0 and out["scaling_like_for_like"] > 0


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
    assert out["roofline"]["pass_order"] == "interleaved" and first["value_slice_order"] > 0 and first["slice_order"]["us_per_query_per_gpu"] > 0
    # Infinity-Cache-sized shards: there is no HBM rate to state, neither as dispatched nor in slice order
    assert out["roofline"]["frac"] is None and first["slice_order"]["frac"] is None and first["slice_order"]["mall_resident"] is True
    assert first["slice_order"]["frac_algorithmic_equiv"] is None and out["roofline"]["achieved"] > 0
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
    # `frac` is a RATE of bytes that really move through HBM; at this Infinity-Cache-sized database there is none to state (null), the
    # kernel's consumption rate stays in `achieved` / `frac_moved` and SURVEY 8(d)'s algorithmic-bytes figure in `frac_algorithmic_equiv`
    assert roof["frac"] is None and roof["mall_resident"] is True and roof["on_die"] is True and "null" in roof["frac_is"] and "traffic" in roof
    assert roof["frac_moved"] is None and roof["frac_algorithmic_equiv"] is None and roof["achieved"] > 0 and roof["achieved_algorithmic_equiv"] >= roof["achieved"]
    assert 0.5 < roof["moved_over_algorithmic"] <= 1.0 and roof["traffic_over_algorithmic"] > 0
    # HBM traffic per launch is MEASURED in the run (two child runs of the timed loop under rocprofv3 --pmc): this Infinity-Cache-sized
    # database is walked in the interleaved order (its 32 passes share one stream of it on die), so anything between 1/32 of the layout
    # bytes plus the queries and a little above all of them
    assert isinstance(roof["traffic"], int) and roof["traffic"] > 0 and "measured in this run" in roof["traffic_source"]
    assert 0.01 < roof["traffic_over_moved_bytes"] < 1.2 and roof["pass_order"] == "interleaved"
    # the bytes really moved (the resident layout is tighter than the reference packing), against spec and against a live read-only probe
    assert roof["moved_GBps"] == roof["achieved"] and roof["read_ceiling_GBps"] > 1000 and roof["frac_vs_read_ceiling"] is None
    assert d["ranks"] == 1 and d["backend"] is None
    assert abs(d["value"] - d["config"]["queries_per_step"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.01
    assert d["headline_check"]["responses_bit_exact_vs_64bit_sums"] is True and d["headline_check"]["unit_queries"] >= 2
    vs = d["value_samples"]  # the contract's region + the same K steps twice more: how much one sample moves
    assert vs["queries_per_sec"][0] == d["value"] and len(vs["queries_per_sec"]) == 3 and vs["min"] <= vs["median"] <= vs["max"]
    cpu = d["cpu_baseline"]
    assert cpu["kind"] in ("reference", "port") and cpu["cores"] >= 1 and cpu["value"] > 0 and isinstance(cpu["sample"], str)
    assert cpu["gpu_results_bit_exact"] is True and cpu["queries_compared"] == 32
    # comparable from box to box: several samples per thread placement, the CPU's name and what the process may run on
    assert cpu["min"] <= cpu["median"] <= cpu["max"] and cpu["value"] == cpu["median"] and isinstance(cpu["cpu_model"], str) and cpu["affinity_mask"]
    for placement in ("threads_bound", "threads_spread", "threads_unbound"):
        assert len(cpu[placement]["samples"]) >= 3 and cpu[placement]["min"] <= cpu[placement]["median"] <= cpu[placement]["max"]


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


def test_default_line_carries_the_other_configs_and_the_cpu_figure_of_the_cpu_config():
    """the default (N = 1) line: `other_configs` -- the headline's loop at other BASELINE shapes inside the same invocation, each with its
    launch time, its rate and its responses checked against exact 64-bit sums -- and the CPU port's figure at BASELINE configs[0]
    (2^16 keys: the reference's own CPU-runnable case).  Run here with small stand-ins for cfg3 / cfg4 / cfg5 so that it takes seconds."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "tiny", "--steps", "2", "--warmup", "1", "--no-setup", "--no-setup-kv",
           "--no-host-path", "--no-live-traffic", "--no-read-ceiling", "--cpu-seconds", "1", "--other-configs", "cfg1,tiny", "--other-steps", "3"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert set(d["other_configs"]) == {"cfg1", "tiny"}
    for name, sec in d["other_configs"].items():
        assert "error" not in sec, sec
        assert sec["value"] > 0 and sec["launch_us"] > 0 and sec["steps"] == 3 and sec["passes_per_launch"] == 32 and sec["pass_order"] == "slice"
        assert sec["responses_bit_exact_vs_64bit_sums"] is True and sec["check"]["unit_queries_ok"] and sec["check"]["dense_and_all_ones_ok"]
        assert sec["frac"] is None and sec["frac_moved"] is None and sec["mall_resident"] is True and sec["moved_GBps_per_gpu"] > 0  # (cache-sized stand-ins: no HBM rate)
        assert abs(sec["value"] - sec["queries_per_step"] / (sec["ms_per_step"] * 1e-3)) / sec["value"] < 0.01
    small = d["cpu_baseline_cfg1"]
    assert small["gpu_results_bit_exact"] is True and small["queries_compared"] == 32 and small["value"] > 0 and "cfg1" in small["workload"]


def test_multirank_line_carries_baselines_multi_gpu_configs_and_the_group_handle():
    """N > 1: behind the headline, the line gains `baseline_multi_gpu_configs` (BASELINE.json's own multi-GPU configs sharded over the ranks,
    slice order, each with `multirank_bit_exact` and rank 0's single-GPU run of the same database) and -- after the process group is gone --
    `respond_host_path_group`: ONE in-process handle over all the devices, the path rust/server_hip.rs gives a drop-in caller.  Three ranks
    on the one GPU under the gloo hook, small stand-ins for cfg4 / cfg5."""
    env = dict(os.environ, CPIR_BENCH_BACKEND="gloo", CPIR_BENCH_SHARE_DEVICE="1", OMP_NUM_THREADS="4")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "4", "--warmup", "2", "--config", "cfg1", "--no-setup",
           "--other-configs", "tiny,cfg1", "--other-steps", "3", "--group-after-ranks", "always"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    first, out = lines[0], lines[-1]
    assert first["baseline_multi_gpu_configs_pending"] is True and first["respond_host_path_group_pending"] is True and first["value"] > 0
    assert not any(k.endswith("_pending") for k in out) and "stage_timed_out" not in out
    secs = out["baseline_multi_gpu_configs"]
    assert set(secs) == {"tiny", "cfg1"}
    for name, sec in secs.items():
        assert "error" not in sec, sec
        assert sec["n_gpus"] == 3 and sec["queries_per_step"] == 96 and sec["pass_order"] == "slice" and sec["value"] > 0
        assert sec["multirank_bit_exact"] is True and sec["ranks_seen"] == [0, 1, 2] and sum(sec["check"]["shard_slots"]) > 0
        ref = sec["single_gpu_reference"]
        assert ref["queries_per_sec"] > 0 and abs(sec["scaling_like_for_like"] - sec["value"] / ref["queries_per_sec"]) < 2e-3
        assert sec["moved_GBps_per_gpu"] > 0 and sec["launch_us"] > 0
    grp = out["respond_host_path_group"]
    assert "error" not in grp, grp
    assert len(grp["shards"]) == 3 and grp["responses_equal_single_device"] is True and grp["device_queries_equal_single_device"] is True
    assert grp["one_caller_us_per_query"] > 0 and grp["eight_callers_queries_per_sec"] > 0 and grp["device_queries_us_per_query"] > 0
    # ... and the group exactly as the Rust drop-in builds it: Server::setup from the KEY-VALUE database over the devices (cpir_server_setup_kv_multi)
    kv = grp["from_kv_database"]
    assert "error" not in kv, kv
    assert kv["entry_point"] == "cpir_server_setup_kv_multi" and len(kv["shards"]) == 3 and kv["server_setup_kv_multi_wall_sec"] > 0
    assert kv["same_filter_params_as_single_device"] and kv["same_hint_as_single_device"] and kv["responses_equal_single_device"] is True
    assert kv["slots_served"] == 65536 and kv["slots_of"] == 77824 and kv["one_caller_us_per_query"] > 0 and kv["eight_callers_queries_per_sec"] > 0
    assert out["value"] == first["value"] and out["multirank_bit_exact"] is True and out["single_gpu_reference"]["queries_per_sec"] > 0


def test_a_stage_that_hangs_ends_the_run_with_a_line_that_says_which():
    """every stage behind the headline has a deadline of its own: ranks that never come out of `baseline_multi_gpu_configs` (a collective
    that does not come back) end the run with exit code 3, and the LAST line names the stage and still carries the headline's figures"""
    env = dict(os.environ, CPIR_BENCH_BACKEND="gloo", CPIR_BENCH_SHARE_DEVICE="1", OMP_NUM_THREADS="4", CPIR_BENCH_TEST_HANG_STAGE="baseline_multi_gpu_configs")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--config", "tiny", "--no-setup",
           "--other-configs", "tiny", "--other-deadline", "5", "--group-after-ranks", "never"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0, p.stdout[-2000:]
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert lines[0]["baseline_multi_gpu_configs_pending"] is True and lines[0]["value"] > 0
    assert lines[-1]["stage_timed_out"].startswith("baseline_multi_gpu_configs") and lines[-1]["value"] == lines[0]["value"]
    assert lines[-1]["multirank_bit_exact"] is True and "single_gpu_reference" in lines[-1]  # what was done before the hang is in the line
