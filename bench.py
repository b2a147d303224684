#!/usr/bin/env python3
"""bench.py -- headline benchmark of the ChalametPIR server hot path on MI355X.

Metric (BASELINE.json): `server_respond` queries/sec + achieved HBM GB/s on the 2^20-key x 1 kB database (3-wise filter:
N = 1 179 648 slots, C = 940 columns, b = 9 bits, 3 fields per packed u32), and `server_setup` wall seconds.

    python bench.py --gpus N --steps K --warmup W

A "step" is one batch of `--queries-per-step` (default: 32 per GPU) distinct synthetic queries; every query is an independent
Server::respond: one full pass of the respond kernel over the packed database resident in HBM (inputs already in HBM
when the timed region starts).  For N > 1 (one process per GPU, launched by torch.distributed.run) the database is
sharded along the filter-slot axis, every rank streams its shard, and the per-shard partial responses of the step's
queries are sum-reduced with ONE RCCL all-reduce per step (u32 wrap-around, bit-exact for any order).  The DATABASE is fixed
and split N ways while the step's batch grows with N (32 queries per GPU), so the bytes a GPU streams per step are the same at
every N => "scaling": "weak" (per-GPU work fixed; `scaling_note` in the line spells it out).  Every N > 1 line carries its own
proof of bit-exactness (`multirank_check`: unit, dense and all-ones queries against the counter-based generator and exact 64-bit
sums, after the timed region; on by default).

`roofline.traffic` (HBM bytes per launch from the PMC counters) is measured in the run itself: the timed loop alone, twice, as a child
process under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (live_traffic() below).

Rank 0 prints ONE JSON line -- in a multi-rank run that line several times: first as soon as the timed respond region and its reduction are
done (with "..._pending": true for what is still to come), then again after every stage behind it -- rank 0's single-GPU reference, BASELINE.json's
own multi-GPU configs (cfg4, cfg5) sharded over the ranks (`baseline_multi_gpu_configs`), the sharded setup, and, once the process group is gone,
the in-process GROUP handle over all visible devices timed in a child of rank 0 (`respond_host_path_group`: the multi-GPU path a drop-in
caller gets) -- each stage under a deadline of its own, so that an optional extra can never cost the headline; take the LAST line.
The default (N = 1, cfg2) line also carries `other_configs` (cfg3 / cfg4 / cfg5: 5 steps x 32 passes each, responses checked against exact
64-bit sums) and `cpu_baseline_cfg1`.  `roofline.frac` is a RATE (bytes really moved / time / 8 TB/s; null where the bytes are served on die);
SURVEY.md 8(d)'s algorithmic-bytes figure is `roofline.frac_algorithmic_equiv`.  `roofline` describes the respond kernel against the HBM roof (8 TB/s,
/opt/skills/guides/MI355X_MICROARCH.md); `cpu_baseline` is the test oracle's restatement of the reference CPU path
(oracle/, kind "port": the Rust reference cannot be built in this image) timed on this box's host cores on the same
database and queries, and it doubles as a full-size bit-exact parity check of the GPU results.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip table); 6290 GB/s is the measured copy ceiling

CONFIGS = {
    # name: (n keys, arity, value bytes)      -- BASELINE.json configs[1..4]
    "cfg2": (1 << 20, 3, 1024),
    "cfg3": (1 << 20, 4, 1024),
    "cfg4": (1 << 22, 3, 1024),
    "cfg5": (1 << 20, 3, 8192),
    "cfg1": (1 << 16, 3, 1024),
    "tiny": (1 << 12, 3, 64),
    # beyond BASELINE.json: a key count at which the reference's bit-length rule (server.rs:193-218) drops to b = 8 (byte-per-field planar)
    "b8": (1 << 23, 3, 256),
}

SEED_D, SEED_Q = 0xD, 0x1000
SETUP_DEADLINE_EXIT_CODE = 3  # a multi-rank run whose sharded-setup extra hung: the flagged line is out, the exit code says it was not a clean run
SEED_MU = bytes(range(32))


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main() -> int:
    # the CPU baseline shares this process: idle OpenMP workers must sleep, not spin (spinning burns the container's CPU
    # quota and throttles the timed loop); has to be in the environment before libgomp is loaded (torch loads it)
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--queries-per-step", type=int, default=0,
                    help="queries in a step's batch; 0 (default) = 32 per GPU: the database is fixed, so a rank's share of a step shrinks with the "
                         "number of GPUs (0.37 ms for 32 queries on an eighth of the headline database) and a step of a fixed 32 queries would "
                         "be timed against the latency of its own barrier")
    ap.add_argument("--query-pool", type=int, default=0, help="distinct queries cycled through (no query-side caching); 0 = twice the step's batch")
    ap.add_argument("--no-setup", action="store_true", help="skip the server_setup timing (needs A: 8.4 GB at cfg2, ~10 s of host XOF)")
    ap.add_argument("--setup-kv", action="store_true",
                    help="time the FULL Server::setup(seed, kv database) incl. filter construction and row encoding -- the reference's own "
                         "`server_setup` bench -- and serve that REAL encoded database (`real_db`).  On by default where the synthetic key-value "
                         "database fits 2 GB of host memory (cfg1-cfg3: ~1.1 GB at cfg2); this flag forces it at the larger configs")
    ap.add_argument("--no-setup-kv", action="store_true", help="skip the key-value setup and the real-database section")
    ap.add_argument("--real-db-keys", type=int, default=4,
                    help="keys looked up end to end on the real database (oracle's client restatement: query -> GPU respond on wire bytes -> decode); "
                         "0 skips the check (it expands the 8.4 GB public matrix on the host once more, ~10 s)")
    ap.add_argument("--setup-deadline", type=float, default=120.0,
                    help="multi-rank runs: seconds the sharded server_setup timing may take after the respond line has been printed")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-read-ceiling", action="store_true", help="skip the live read-only-stream probe (roofline.read_ceiling_GBps)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure roofline.traffic in this run (two child runs of the timed loop under rocprofv3 --pmc); quote the committed pass instead")
    ap.add_argument("--headline-only", action="store_true", help=argparse.SUPPRESS)  # internal: the timed loop and nothing else (the counter passes' child)
    ap.add_argument("--no-host-path", action="store_true", help="skip timing Server.respond on host buffers (PCIe inclusive)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="time budget of the CPU baseline (a third of it per thread placement, three samples each)")
    ap.add_argument("--cpu-baseline-child", default="", help=argparse.SUPPRESS)  # internal: the CPU baseline's own process (no torch, no GPU)
    ap.add_argument("--tune", default="", help="comma list key=value for cpir_tuning_set, e.g. respond.interleave_passes=0")
    ap.add_argument("--sweep", action="store_true", help="time every respond kernel variant (stderr table) before the run")
    ap.add_argument("--enqueue", default="batch", choices=["batch", "python"],
                    help="how a step's launches are enqueued: one C call for the step (default) or one ctypes call per query")
    ap.add_argument("--group-shards", type=int, default=0,
                    help="also time Server::respond on host buffers through an in-process GROUP handle (cpir_server_setup_multi): 0 = one "
                         "shard per visible device when there are at least two, K = K shards cycled over the visible devices")
    ap.add_argument("--group-child", action="store_true", help=argparse.SUPPRESS)  # internal: the group timing runs in a child process
    ap.add_argument("--other-configs", default="auto",
                    help="other BASELINE.json configs run the headline's way inside this invocation: 'auto' (default; with --config cfg2: cfg3,cfg4,cfg5 on "
                         "one GPU -> `other_configs`; cfg4,cfg5 sharded over the ranks of a multi-GPU run -> `baseline_multi_gpu_configs`), 'none', or a "
                         "comma list of config names")
    ap.add_argument("--other-steps", type=int, default=5, help="timed steps of each of those sections (2 warm-up steps in front)")
    ap.add_argument("--other-deadline", type=float, default=240.0, help="multi-rank runs: seconds each of those sections may take")
    ap.add_argument("--no-cpu-cfg1", action="store_true", help="skip the CPU baseline of BASELINE.json configs[0] (2^16 keys, the CPU reference path) in the default line")
    ap.add_argument("--group-after-ranks", default="auto", choices=["auto", "always", "never"],
                    help="multi-rank runs: after the process group is gone, rank 0 times the in-process GROUP handle over all visible devices in a child "
                         "process (`respond_host_path_group`): auto = with --config cfg2")
    ap.add_argument("--shard-of", type=int, default=0,
                    help="tuning aid: run ONE process on rank 0's shard of a K-way split (no collective); the JSON line then "
                         "describes that shard's kernel only")
    ap.add_argument("--no-multirank-check", action="store_true",
                    help="N > 1: skip the default-on bit-exactness check behind the timed region (unit / dense / all-ones queries against the "
                         "counter-based generator and exact 64-bit sums; size-independent, runs at every config)")
    ap.add_argument("--no-like-for-like", action="store_true",
                    help="N > 1: skip the second timed loop in slice order (`value_slice_order`) and rank 0's single-GPU run of the whole database "
                         "(`single_gpu_reference`) that make `scaling_like_for_like` -- the scaling figure free of on-die reuse")
    ap.add_argument("--verify", action="store_true",
                    help="rank 0 re-derives the step's responses with the CPU oracle from the full synthetic DB (small configs only)")
    args = ap.parse_args()

    # `python bench.py --gpus N` with no launcher around it: start the N ranks ourselves, as a CHILD process, before anything in this
    # process has touched the GPU (no torch import yet, no HIP call) -- never exec, never re-launch later.  Under
    # torch.distributed.run (WORLD_SIZE set) this is skipped and the process is one rank.
    if args.cpu_baseline_child:
        return cpu_baseline_child(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.group_child and args.shard_of <= 1:
        return launch_ranks(args.gpus)

    import torch
    import torch.distributed as dist

    import chalametpir_amd as cp
    from chalametpir_amd.distributed import ShardedServer, shard_range

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if args.queries_per_step <= 0:
        args.queries_per_step = 32 * max(1, world)
    if args.query_pool <= 0:
        args.query_pool = 2 * args.queries_per_step
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}; using WORLD_SIZE")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    # test hooks: CPIR_BENCH_SHARE_DEVICE=1 puts every rank on GPU 0 and CPIR_BENCH_BACKEND=gloo swaps the collective, so the
    # multi-rank path can be exercised on a one-GPU box (RCCL refuses two ranks on one device); never used for reported numbers
    if os.environ.get("CPIR_BENCH_SHARE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("CPIR_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend)
    device = cp.Device(local_rank)
    for kv in filter(None, args.tune.split(",")):
        k, v = kv.split("=")
        cp.tuning_set(k, int(v))

    if args.group_child:
        return group_child(cp, torch, device, args)

    n_keys, arity, value_bytes = CONFIGS[args.config]
    b = cp.find_encoded_db_matrix_element_bit_length(n_keys)
    _, _, N = cp.filter_shape(arity, n_keys)
    C = cp.encoded_num_cols(value_bytes, b)
    cf = 2 if b >= 11 else (3 if b >= 9 else 4)
    W = -(-N // cf)
    mask = (1 << b) - 1
    stream = torch.cuda.current_stream()
    full_layout = cp.dtc_layout_for(N, C, b)  # packing of the resident DB (dense64 where offered)

    # ---- this rank's shard of the synthetic encoded DB, generated in HBM, packed, and D freed ---------------------------
    lo, hi = shard_range(N, full_layout, rank, world)
    if args.shard_of > 1 and world == 1:
        lo, hi = shard_range(N, full_layout, 0, args.shard_of)
    t0 = time.time()
    D_dev = torch.empty(((hi - lo), C), dtype=torch.int32, device="cuda")
    if hi > lo:
        device.synth_fill(D_dev, (hi - lo) * C, SEED_D, index0=lo * C, mask=mask, stream=stream)
    sharded = ShardedServer.from_device_matrix(D_dev, lo, hi, C, b, N, device, stream=stream)
    torch.cuda.synchronize()
    del D_dev
    torch.cuda.empty_cache()
    pack_seconds = time.time() - t0

    pool = args.query_pool
    q_pool = torch.empty((pool, N), dtype=torch.int32, device="cuda")
    for i in range(pool):
        device.synth_fill(q_pool, N, SEED_Q + i, offset_words=i * N, stream=stream)
    qps_step = args.queries_per_step
    torch.cuda.synchronize()

    mem_peak = [0]

    def mem_mark():
        """device memory in use on this rank's card right now (whole card: hipMemGetInfo), highest value seen kept"""
        free, total = torch.cuda.mem_get_info()
        mem_peak[0] = max(mem_peak[0], total - free)

    mem_mark()
    if pool % qps_step != 0:
        raise SystemExit("--query-pool must be a multiple of --queries-per-step")
    # every query of a step is an independent pass over the database (batch_fusion = 0): the batch entry point is only
    # used to enqueue the step's launches from C instead of one Python/ctypes round trip per query
    cp.tuning_set("respond.batch_fusion", 0)
    loop = RespondLoop(torch, dist, sharded, q_pool, qps_step, C, world, stream, args.enqueue)
    run_step, drain, timed_region, step_counter, r_step = loop.run_step, loop.drain, loop.timed_region, loop.step_counter, loop.r_bufs[0]

    if args.sweep and rank == 0:
        sweep(cp, torch, run_step, qps_step)

    elapsed, kernel_region_ms = timed_region(args.warmup, args.steps)
    # the contract's figure is THAT region (W warm-up steps, exactly K steps); two more regions of K steps right behind it say how much one such
    # sample moves on this box (`value_samples`: the driver's 20 steps are 0.12 s of device time)
    repeats = []
    if world == 1 and not args.headline_only and args.shard_of <= 1:
        for _ in range(2):
            e2, _ms = timed_region(0, args.steps)
            repeats.append(args.steps * qps_step / e2)

    n_queries = args.steps * qps_step
    qps = n_queries / elapsed
    full_bytes = 4 * C * W + 4 * N + 4 * C
    # ONE respond launch answers the step's queries as that many independent passes over the database (enqueue=batch), or one
    # query (enqueue=python); bytes and duration below are per LAUNCH, as the kernel trace sees them
    limit_mb = next((int(kv.split("=")[1]) for kv in args.tune.split(",") if kv.startswith("respond.multi_pass_limit_mb=")), 2560)
    big = (sharded.local is not None and full_layout.packing != 2 and
           int(sharded.local.layout.total_words) * 4 > (limit_mb << 20))  # one launch per query there (VALU kernels only)
    passes_per_launch = qps_step if (args.enqueue == "batch" and not big) else 1
    shard_words = int(sharded.local.layout.total_words) if sharded.local is not None else 0
    roof = respond_roofline(C, cf, b, hi - lo, full_layout, shard_words, passes_per_launch, kernel_region_ms * 1e3 / n_queries * passes_per_launch)
    launch_bytes, launch_bytes_q, moved_bytes = roof["bytes_per_launch"], roof["bytes_per_launch"] // passes_per_launch, roof["moved_bytes_per_launch"]
    packing = {0: "reference", 1: "dense64", 2: "planar"}[int(full_layout.packing)]

    result = {
        "metric": "server_respond_queries_per_sec",
        "value": round(qps, 2),
        "unit": "queries/s",
        "n_gpus": world,
        "ranks": dist.get_world_size() if world > 1 else 1,  # what the process group itself reports
        "backend": (dist.get_backend() if world > 1 else None),  # "nccl" = RCCL over xGMI on ROCm
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        # per-GPU work per step is fixed as N grows: 32 queries per GPU x that GPU's 1/N of the (fixed) database = 32 database passes'
        # worth of bytes per GPU and step at every N -- the contract's "weak"; the database itself does not grow with N
        "scaling": "weak",
        "scaling_note": "fixed database split N ways along the filter slots, 32 queries per GPU and step: every GPU streams the same bytes per step "
                        "at every N; value = all queries of all ranks / max-over-ranks wall time",
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic",
        "config": {
            "workload": workload_name(args.config, N, C, b, cf),
            "queries_per_step": qps_step,
            "query_pool": pool,
            "sharding": f"N split over {world} GPU(s), one all-reduce of {qps_step}x{C} u32 per step, overlapped with the next step" if world > 1 else "single GPU",
        },
        "achieved_hbm_GBps_whole_job": round(full_bytes * qps / 1e9, 1),
        "algorithmic_bytes_per_query": full_bytes,
        "roofline": roof,
        "pack_seconds": round(pack_seconds, 3),
    }
    if repeats:
        samples = [qps] + repeats
        result["value_samples"] = {"queries_per_sec": [round(x, 2) for x in samples], "min": round(min(samples), 2), "median": round(sorted(samples)[1], 2),
                                   "max": round(max(samples), 2), "note": "`value` is the first: the contract's timed region; the other two are the same K steps again, no warm-up between"}
    if world > 1 and os.environ.get("CPIR_BENCH_SHARE_DEVICE") == "1":
        # the one-GPU rehearsal hook: every rank on GPU 0, gloo as the collective -- plumbing only, no figure of this line is a measurement
        result["rehearsal_ranks_share_one_device"] = True
    # LIKE FOR LIKE (N > 1, and the one-shard tuning runs): a shard below ~1 GB runs its passes in the interleaved order -- concurrent passes
    # share database bytes on die --, the N = 1 headline streams the database from HBM for every query.  `value` is what the product
    # dispatches; `value_slice_order` is the same shards, the same steps, every pass its own stream (slice order, `nt` loads), i.e. the
    # figure that may be divided by the N = 1 headline: `scaling_like_for_like` (filled in below, against rank 0's own single-GPU run of
    # the whole database in this very process).
    if (world > 1 or args.shard_of > 1) and not args.no_like_for_like:
        roof0 = result["roofline"]
        if roof0["pass_order"] == "interleaved":
            user_order = next((int(kv.split("=")[1]) for kv in args.tune.split(",") if kv.startswith("respond.interleave_passes=")), -1)
            cp.tuning_set("respond.interleave_passes", 0)
            elapsed_s, region_ms_s = timed_region(max(1, min(args.warmup, 3)), args.steps)
            cp.tuning_set("respond.interleave_passes", user_order)
        else:
            elapsed_s, region_ms_s = elapsed, kernel_region_ms
        launch_us_s = region_ms_s * 1e3 / n_queries * passes_per_launch
        result["value_is"] = (("the product's dispatch at this shard size: passes in the INTERLEAVED order, i.e. concurrent passes share database bytes on "
                               "die -- not comparable with the N = 1 headline; compare `value_slice_order` (same shards, every pass its own stream) with "
                               "it: `scaling_like_for_like`") if roof0["pass_order"] == "interleaved" else
                              "slice order: every pass its own stream of the shard from HBM, as the N = 1 headline (value_slice_order = value)")
        result["value_slice_order"] = round(n_queries / elapsed_s, 2)
        roof_s = respond_roofline(C, cf, b, hi - lo, full_layout, shard_words, passes_per_launch, launch_us_s, force_slice=True)
        result["slice_order"] = {
            "queries_per_sec": round(n_queries / elapsed_s, 2),
            "ms_per_step": round(elapsed_s / args.steps * 1e3, 4),
            "us_per_query_per_gpu": round(region_ms_s * 1e3 / n_queries, 2),
            "frac": roof_s["frac"],  # bytes really moved per GPU / launch time / 8 TB/s (null where a pass fits the Infinity Cache)
            "frac_moved": roof_s["frac_moved"],
            "frac_algorithmic_equiv": roof_s["frac_algorithmic_equiv"],
            "mall_resident": roof_s["mall_resident"],
            "note": "the same shards and steps with every pass its own stream of the shard (slice order, nt loads): comparable with the N = 1 "
                    "headline; `value` is the product's dispatch (pass_order above), which at this shard size shares database bytes on die",
        }
    if world == 1 and not args.headline_only and args.shard_of <= 1 and not args.no_multirank_check:
        # the N = 1 line's own proof (as every other config of the line has one): unit / dense / all-ones queries against exact 64-bit sums of
        # the regenerated rows, both dispatch modes -- BEHIND the timed regions (a check in front of them would also warm the device up:
        # `value_samples` shows that the first region of a process runs ~2 % below the next ones; the contract's W warm-up steps stay all there is)
        drain()
        try:
            chk1 = multirank_check(torch, dist, device, sharded, N, C, b, mask, lo, hi, 0, 1, full_layout, stream)
            result["headline_check"] = {"responses_bit_exact_vs_64bit_sums": chk1.get("multirank_bit_exact"), "unit_queries": chk1.get("unit_queries"),
                                        "dense_and_all_ones_queries": chk1.get("dense_and_all_ones_queries"), "dispatch_modes": chk1.get("dispatch_modes"),
                                        "seconds": chk1.get("seconds")}
        except Exception as exc:  # noqa: BLE001 -- a check that could not run is reported as such
            log(f"headline check failed to run: {exc!r}")
            result["headline_check"] = {"responses_bit_exact_vs_64bit_sums": None, "error": repr(exc)}
        cp.tuning_set("respond.batch_fusion", 0)
    if world > 1 and not args.no_multirank_check:
        drain()
        try:
            chk = multirank_check(torch, dist, device, sharded, N, C, b, mask, lo, hi, rank, world, full_layout, stream)
        except Exception as exc:  # noqa: BLE001 -- the check must never cost the line; a check that could not run is reported as such
            log(f"rank {rank}: multirank check failed to run: {exc!r}")
            chk = {"multirank_bit_exact": None, "error": repr(exc)}
        mem_mark()
        result["device_bytes_in_use_peak_seen"] = mem_peak[0]  # this rank's card (hipMemGetInfo), sampled after the pools and after the check
        result["multirank_bit_exact"] = chk.get("multirank_bit_exact")
        result["ranks_seen"] = chk.get("ranks_seen")
        result["multirank_check"] = chk
    if args.shard_of > 1 and world == 1:
        result["config"]["sharding"] = f"TUNING RUN: rank 0's shard of a {args.shard_of}-way split, alone, no collective"
    # `frac` is a rate (respond_roofline): the bytes the kernel really moves over the launch time against the 8 TB/s spec; beside it what a bare
    # read-only stream reaches on THIS device (`frac_vs_read_ceiling`, measured just now by the tool built from scripts/hbm_read_ceiling.hip,
    # in a child process after the timed region) and SURVEY.md 8(d)'s algorithmic-bytes figure (`frac_algorithmic_equiv`).
    roof = result["roofline"]
    roof["read_ceiling_GBps"] = None
    roof["frac_vs_read_ceiling"] = None
    # (a child process; this one waits for it with an idle device, so it runs behind the sections that continue the headline's loop)
    def measure_read_ceiling():
        if rank == 0 and not args.no_read_ceiling:
            ceil = read_ceiling(min(max(4 * shard_words, 256 << 20), 4 << 30))
            if ceil:
                roof["read_ceiling_GBps"] = ceil["read_ceiling_GBps"]
                roof["frac_vs_read_ceiling"] = None if roof["on_die"] else round(roof["moved_GBps"] / ceil["read_ceiling_GBps"], 4)
                roof["read_ceiling_note"] = f"{ceil['kernel']}; {ceil['buffer_bytes'] / 1e9:.2f} GB read once per launch, best of several launch shapes"

    # (run AFTER every timed section of this process: a rocprofv3 counter pass leaves the device at the profiling power state for a
    # while -- the fused batches, which are sensitive to the matrix cores' clock, measured 13.1 us per query right behind the counter
    # passes and 10.7-11.0 without them in the same build on the same box)
    def measure_traffic():
        # HBM traffic per launch cannot be sampled from inside this process (PMC counters need rocprofv3 around it): the last committed
        # counter pass for this exact workload and packing (scripts/profile_gpu.sh -> profiles/respond_traffic.json) is quoted, with the
        # commit it was taken at and whether the kernel source has changed since.
        if world == 1 and not args.headline_only:
            tr = committed_traffic(args.config, launch_bytes_q, packing)
            live = None if args.no_live_traffic else live_traffic(args, passes_per_launch)
            if live:
                # measured NOW: the same timed loop in two child processes under rocprofv3, one counter each (the pool refuses --pmc next to
                # other trace domains, and FETCH_SIZE / WRITE_SIZE do not fit one pass), corrected as the guide's HBM section prescribes
                roof["traffic"] = int(live["bytes_per_launch"])
                roof["traffic_source"] = live["source"]
                roof["traffic_over_moved_bytes"] = round(live["bytes_per_launch"] / moved_bytes, 4) if moved_bytes else None
                roof["traffic_over_algorithmic"] = round(live["bytes_per_launch"] / launch_bytes, 4) if launch_bytes else None
                if tr:
                    roof["traffic_committed_pass"] = int(tr["traffic_bytes_per_pass"] * passes_per_launch)  # profiles/respond_traffic.json, for comparison
            elif tr:
                roof["traffic"] = int(tr["traffic_bytes_per_pass"] * passes_per_launch)
                roof["traffic_over_algorithmic"] = round(roof["traffic"] / launch_bytes, 4) if launch_bytes else None
                roof["traffic_source"] = ("profiles/respond_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate passes; "
                                          f"NOT measured in this run) taken at commit {tr.get('git_head', 'unknown')}")
                roof["traffic_kernel_source_unchanged"] = tr.get("kernel_source_sha256") == kernel_source_sha256()

    # row f3: the same queries answered 4 per pass over the database (fused batch kernel) -- reported beside the
    # headline, never as the headline: the headline streams the whole database for every single query
    if world == 1 and not args.headline_only:
        cp.tuning_set("respond.batch_fusion", 1)
        nbq = 48 if pool >= 64 else qps_step  # 48 queries a launch where the pool allows it (default: 48 of 64)
        per_pass = cp.respond_batch_pass_width(full_layout, nbq)  # planar: 2 wide passes of 24 (the library says how it cuts the batch)
        rb = torch.zeros((nbq, C), dtype=torch.int32, device="cuda")

        def fused_step(k):
            off = (16 * k) % (pool - nbq + 1) if pool > nbq else 0
            sharded.respond_partial_device(q_pool[off:off + nbq], rb, batch=nbq, stream=stream)

        # (a dozen warm-up launches, not three: measured behind a child process that left this one's device idle for a second or two, the
        # wide pass -- sensitive to the matrix cores' clock -- took 13.1-13.7 us per query after 2 ms of warm-up, 10.7-11.0 straight behind
        # the headline loop, same build, same box; the read-ceiling child now runs after these sections)
        for k in range(24):
            fused_step(k)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        n_fused_steps = max(8, args.steps // 2)
        for k in range(n_fused_steps):
            fused_step(k)
        e1.record(stream)
        torch.cuda.synchronize()
        fused_us = e0.elapsed_time(e1) * 1e3 / (n_fused_steps * nbq)
        result["batched_respond"] = {
            "queries_per_pass": per_pass,
            "queries_per_launch": nbq,
            "queries_per_sec": round(1e6 / fused_us, 1),
            "us_per_query": round(fused_us, 2),
            "note": "cpir_server_respond_batch_device with batch fusion: the queries of a pass share one stream of the packed DB (planar: up to 24 per "
                    "pass -- the wide pass, six row sets of the i8 matrix cores walked by one 8-wave block per CU; 12 per pass up to round 4's "
                    "step-major kernel alone); same results bit for bit",
        }
        del rb
        cp.tuning_set("respond.batch_fusion", 0)
    # one query per LAUNCH (what a caller of cpir_server_respond_device pays when it has a single query): launches back to back
    if world == 1 and not args.headline_only:
        r1 = torch.empty(C, dtype=torch.int32, device="cuda")
        for i in range(4):
            sharded.local.respond_device(q_pool[i % pool], r1, stream=stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        n_lone = 64
        for i in range(n_lone):
            sharded.local.respond_device(q_pool[i % pool], r1, stream=stream)
        e1.record(stream)
        torch.cuda.synchronize()
        lone_us = e0.elapsed_time(e1) * 1e3 / n_lone
        result["lone_query_launch"] = {
            "us_per_query": round(lone_us, 2),
            "queries_per_sec": round(1e6 / lone_us, 1),
            "note": "one query per launch, launches back to back on one stream (r zeroed by a small kernel + one respond kernel each; on the planar packing the "
                    "wide-pass kernel with one row set, which adds the correction terms itself)",
        }
    # also row f3: the same independent passes (one query each, no fusion), but walked in the interleaved order so that concurrent
    # passes share database bytes in L2 / Infinity Cache -- above the HBM roof by construction, hence never the headline
    if world == 1 and full_layout.packing == 2 and passes_per_launch > 1 and not args.headline_only:
        cp.tuning_set("respond.interleave_passes", 1)
        for _ in range(3):
            run_step()
        drain()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        n_c_steps = max(4, args.steps // 4)
        for _ in range(n_c_steps):
            run_step()
        drain()
        e1.record(stream)
        torch.cuda.synchronize()
        c_us = e0.elapsed_time(e1) * 1e3 / (n_c_steps * qps_step)
        result["concurrent_respond"] = {
            "queries_in_flight": qps_step,
            "queries_per_sec": round(1e6 / c_us, 1),
            "us_per_query": round(c_us, 2),
            "equivalent_GBps": round(full_bytes / c_us / 1e3, 1),
            "note": "same launch as the headline (one query per pass), passes walked in the interleaved order with cached loads: "
                    "blocks streaming the same tiles for different queries share them on die, so this is not an HBM-bound number",
        }
        cp.tuning_set("respond.interleave_passes", -1)
    measure_read_ceiling()
    # the C-ABI host path a Rust caller uses: query bytes on the host -> pinned copy -> H2D -> kernel -> D2H -> bytes
    if world == 1 and not args.no_host_path:
        cp.tuning_set("respond.batch_fusion", 1)  # the library's own default: coalesced callers share one stream of the database
        result["respond_host_path"] = host_path_timing(sharded.local, q_pool, N, torch)
        cp.tuning_set("respond.batch_fusion", 0)
    # the same host path from plain C threads (examples/host_respond_bench.c, built by the Makefile): what a caller without an interpreter
    # sees.  A child process with a deadline, after this process's own host-path timing: an optional extra that can never take the
    # headline down with it.
    if world == 1 and not args.no_host_path and args.config in ("cfg2", "cfg3", "cfg5", "cfg1"):
        exe = os.path.join(ROOT, "chalametpir_amd", "lib", "host_respond_bench")
        if os.path.exists(exe):
            import subprocess

            # twice: the synthetic matrix as it is, and with one slot in nine empty (CPIR_BENCH_HOLES=9: what a real 3-wise encoded database
            # looks like to the server -- it leaves those rows out of HBM and every caller compacts its query while staging it).  These are
            # the figures to read for CONCURRENT callers: the Python threads of `respond_host_path` serialise on the interpreter lock
            # (~50-100 us of Python per call: they cannot deliver more than 7-9 k calls/s whatever the library does).
            for key, extra_env in (("respond_host_path_native", {}), ("respond_host_path_native_compacted", {"CPIR_BENCH_HOLES": "9"})):
                try:
                    cp_run = subprocess.run([exe, str(n_keys.bit_length() - 1), str(value_bytes), str(arity)], capture_output=True, text=True, timeout=150,
                                            env=dict(os.environ, **extra_env))
                    if cp_run.returncode == 0:
                        result[key] = json.loads(cp_run.stdout.strip().splitlines()[-1])
                    else:
                        log(f"host_respond_bench failed (rc {cp_run.returncode}): {cp_run.stderr[-300:]}")
                except Exception as exc:  # noqa: BLE001
                    log(f"host_respond_bench skipped: {exc}")
    if world == 1 and not args.no_host_path:
        n_vis = torch.cuda.device_count()
        k = args.group_shards or (n_vis if n_vis >= 2 else 0)
        if k >= 1:
            # in a CHILD process with a deadline: this path drives every visible device from one process, which a one-GPU box
            # cannot rehearse -- an optional extra must never be able to take the headline down with it
            result["respond_host_path_group"] = run_group_child(args, k)
    if args.verify and args.shard_of > 1 and world == 1:
        result["verified_vs_oracle"] = None  # (a lone shard's partial responses are not the database's responses: nothing to compare with)
    elif args.verify:
        drain()
        result["verified_vs_oracle"] = verify(run_step, drain, step_counter, r_step, qps_step, pool, N, C, b, mask, rank, torch)
    # The other single-GPU configs of BASELINE.json in the DEFAULT line (driver-timed evidence for them): 5 steps x 32 passes each, the
    # headline's loop, two dense responses + unit queries against exact 64-bit sums.  Before the CPU baseline and the counter passes.
    others = other_config_names(args, world)
    if world == 1 and others and not args.headline_only:
        result["other_configs"] = {}
        for name in others:
            try:
                result["other_configs"][name] = config_section(cp, torch, dist, device, name, 0, 1, stream, steps=args.other_steps, warmup=2, tune=args.tune)
            except Exception as exc:  # noqa: BLE001 -- an extra must never cost the headline line
                log(f"other config {name} failed: {exc!r}")
                result["other_configs"][name] = {"error": repr(exc)}
                torch.cuda.empty_cache()
    if rank == 0 and world == 1:
        if not args.no_cpu_baseline:
            try:
                result["cpu_baseline"] = cpu_baseline(sharded.local, q_pool, r_step, N, C, b, full_bytes, args.cpu_seconds, torch, stream, args.config)
            except Exception as exc:  # noqa: BLE001 -- the baseline beside the figure must never cost the figure
                log(f"CPU baseline failed: {exc!r}")
                result["cpu_baseline"] = {"error": repr(exc), "kind": "port"}
            if args.config != "cfg1" and not args.headline_only and not args.no_cpu_cfg1:
                # BASELINE.json configs[0] IS the CPU reference path (2^16 keys): its figure from the same port, on the same cores
                try:
                    result["cpu_baseline_cfg1"] = cpu_baseline_small(cp, device, torch, "cfg1", stream)
                except Exception as exc:  # noqa: BLE001
                    log(f"cfg1 CPU baseline failed: {exc!r}")
                    result["cpu_baseline_cfg1"] = {"error": repr(exc)}
        if not args.no_setup:
            result.update(setup_timing(cp, device, torch, sharded, N, C, b, mask, stream))
            result["setup_roofline"] = setup_kernel_roofline(cp, device, torch, N, C, b, cf, mask, full_layout, stream)
        kv_fits = n_keys * (32 + value_bytes) <= (2 << 30)
        if (args.setup_kv or kv_fits) and not args.no_setup_kv and not args.headline_only:
            try:
                result.update(setup_kv_and_real_db(cp, device, torch, args, n_keys, arity, value_bytes, q_pool, N, C, b, cf, stream))
            except Exception as exc:  # noqa: BLE001 -- an extra must never cost the headline line
                log(f"key-value setup / real database section failed: {exc!r}")
                result["server_setup_kv_error"] = repr(exc)
    measure_traffic()

    def add_single_gpu_reference():
        if rank != 0 or "value_slice_order" not in result or "single_gpu_reference" in result:
            return
        try:
            ref = single_gpu_reference(cp, torch, device, ShardedServer, args.steps, args.tune, N, C, b, mask, q_pool, stream)
            result["single_gpu_reference"] = ref
            n_shards = world if world > 1 else args.shard_of
            # (a one-shard tuning run: every query needs the partial of EVERY shard, so the rate at which one GPU answers its 1/N of a query
            # IS the rate the N-GPU job answers whole queries at, exchange aside)
            result["scaling_like_for_like"] = round(result["value_slice_order"] / ref["queries_per_sec"], 3)
            result["scaling_as_dispatched"] = round(result["value"] / ref["queries_per_sec"], 3)
            result["scaling_note_like_for_like"] = (f"{'N' if world > 1 else 'shard-of'} = {n_shards}: value_slice_order / single_gpu_reference.queries_per_sec "
                                                    "(both slice order, nt loads: every query its own stream of the database from HBM)"
                                                    + ("" if world > 1 else "; one shard alone, no exchange"))
        except Exception as exc:  # noqa: BLE001 -- an extra must never cost the line
            log(f"single-GPU reference failed: {exc!r}")
            result["single_gpu_reference"] = {"error": repr(exc)}

    if world == 1:
        add_single_gpu_reference()  # (the one-shard tuning runs)
        print(json.dumps(result), flush=True)
        return 0

    # ---- N > 1: everything behind the headline runs in STAGES, each under a wall-clock deadline ------------------------------------------
    # The headline must not be hostage to an extra: rank 0 prints the respond line NOW and again after every stage (a consumer takes the
    # LAST line).  A stage is a chain of collectives -- an exception can be caught, a hang cannot -- so ONE watchdog thread per rank holds
    # the deadline of the stage in progress.  A run that hits a deadline is NOT a success: rank 0 prints the line once more with
    # "stage_timed_out" (and, for the sharded setup, "server_setup_timed_out": true) -- so the last line says what happened and still
    # carries the measured respond figures -- and every rank exits with code 3: never a restart, never an exec.
    import threading

    pending = {}
    if not args.no_like_for_like:
        pending["single_gpu_reference_pending"] = True
    if others:
        pending["baseline_multi_gpu_configs_pending"] = True
    if not args.no_setup:
        pending["server_setup_pending"] = True
    if group_child_wanted(args):
        pending["respond_host_path_group_pending"] = True

    last_emitted = [None]

    def emit():
        if rank == 0:
            line = json.dumps(dict(result, **pending))
            if line != last_emitted[0]:  # (only when something has changed since the last line)
                print(line, flush=True)
                last_emitted[0] = line

    emit()
    stage = {"label": None, "deadline": None}
    stage_lock = threading.Lock()
    all_done = threading.Event()

    def watchdog():
        while not all_done.wait(0.25):
            with stage_lock:
                label, deadline = stage["label"], stage["deadline"]
            # (rank 0 first: its flagged line must be out before another rank's exit makes the launcher stop everyone)
            if label is None or time.monotonic() < deadline + (0.0 if rank == 0 else 3.0):
                continue
            log(f"rank {rank}: stage '{label}' exceeded its deadline: exit code {SETUP_DEADLINE_EXIT_CODE}; the respond figures of the line stand")
            if rank == 0:
                flags = {"stage_timed_out": label}
                if label == "server_setup":
                    flags.update(server_setup_timed_out=True, server_setup_error=f"deadline of {args.setup_deadline:.0f} s exceeded (a collective did not come back)")
                print(json.dumps(dict(result, **pending, **flags)), flush=True)
            sys.stdout.flush()
            os._exit(SETUP_DEADLINE_EXIT_CODE)

    threading.Thread(target=watchdog, daemon=True).start()

    def run_stage(label, seconds, fn):
        """fn() on every rank under the stage's deadline; an exception on this rank is logged and reported, never raised"""
        with stage_lock:
            stage["label"], stage["deadline"] = label, time.monotonic() + seconds
        try:
            hang = os.environ.get("CPIR_BENCH_TEST_HANG_STAGE")  # test hook: this rank never comes out of the named stage (tests/test_gpu_multirank.py)
            if hang and label.startswith(hang):
                time.sleep(1e6)
            return fn()
        except Exception as exc:  # noqa: BLE001
            log(f"rank {rank}: stage '{label}' failed: {exc!r}")
            return {"error": repr(exc)}
        finally:
            with stage_lock:
                stage["label"] = None

    if "single_gpu_reference_pending" in pending:
        # rank 0 alone (the whole database on its GPU, the N = 1 headline's loop); the others wait at the barrier -- OUTSIDE the setup's deadline
        def ref_stage():
            add_single_gpu_reference()
            dist.barrier()

        run_stage("single_gpu_reference", 180.0, ref_stage)
        pending.pop("single_gpu_reference_pending")
        emit()
    if others:
        # BASELINE.json configs[3] and configs[4] are DEFINED as 8-GPU configs: sharded N ways here, slice order, each with its own proof
        result["baseline_multi_gpu_configs"] = {}
        for name in others:
            sec = run_stage(f"baseline_multi_gpu_configs.{name}", args.other_deadline,
                            lambda name=name: config_section(cp, torch, dist, device, name, rank, world, stream, steps=args.other_steps, warmup=2,
                                                             single_ref=not args.no_like_for_like, tune=args.tune))
            result["baseline_multi_gpu_configs"][name] = sec
        pending.pop("baseline_multi_gpu_configs_pending")
        emit()
    if not args.no_setup:
        def setup_stage():
            if os.environ.get("CPIR_BENCH_TEST_HANG_SETUP") == "1":  # test hook: a collective that never comes back (tests/test_gpu_multirank.py)
                time.sleep(1e6)
            return setup_timing_sharded(cp, device, torch, dist, N, C, b, mask, lo, hi, rank, stream, full_layout)

        extra = run_stage("server_setup", args.setup_deadline, setup_stage)
        if "error" in extra:
            extra = {"server_setup_error": extra["error"]}
        mem_mark()
        if rank == 0:
            result.update(extra)
            result["device_bytes_in_use_peak_seen"] = mem_peak[0]  # rank 0's card, sampled after the pools, the check and the setup
        pending.pop("server_setup_pending")
    # this rank's device memory goes before the teardown: what runs next on rank 0 (the group child) has the devices to itself
    del loop, run_step, drain, timed_region, r_step, sharded, q_pool
    torch.cuda.empty_cache()

    def teardown():
        dist.barrier()
        dist.destroy_process_group()

    run_stage("teardown", 60.0, teardown)
    emit()
    all_done.set()
    if rank == 0 and "respond_host_path_group_pending" in pending:
        # THE OTHER multi-GPU path -- the one a drop-in caller gets (rust/server_hip.rs -> cpir_server_setup_kv_multi: one process, every visible
        # device behind one handle): timed in a CHILD process of rank 0 once the process group is gone and the other ranks have left
        # (they have freed their device memory above and exit right behind the barrier), with a deadline of its own.
        time.sleep(2.0)
        n_vis = torch.cuda.device_count()
        k = args.group_shards or (n_vis if (n_vis >= 2 and os.environ.get("CPIR_BENCH_SHARE_DEVICE") != "1") else world)
        result["respond_host_path_group"] = run_group_child(args, k)
        pending.pop("respond_host_path_group_pending")
        emit()
    return 0


def other_config_names(args, world):
    """the other BASELINE.json configs this invocation also runs (config_section): the default (cfg2) line carries cfg3 / cfg4 / cfg5 on one
    GPU, and cfg4 / cfg5 -- BASELINE's own 8-GPU configs -- sharded over the ranks of a multi-GPU run"""
    if args.other_configs == "none" or args.shard_of > 1:
        return []
    if args.other_configs != "auto":
        names = [n for n in args.other_configs.split(",") if n]
        for n in names:
            if n not in CONFIGS:
                raise SystemExit(f"--other-configs: unknown config {n}")
        return names
    if args.config != "cfg2":
        return []
    return ["cfg3", "cfg4", "cfg5"] if world == 1 else ["cfg4", "cfg5"]


def group_child_wanted(args):
    return (not args.no_host_path) and args.group_after_ranks != "never" and (args.group_after_ranks == "always" or args.config == "cfg2")


def run_group_child(args, shards: int, timeout: float = 240.0):
    """`--group-child` in a child process with a deadline: Server::respond through ONE in-process group handle (cpir_server_setup_multi)
    over `shards` shards cycled over the visible devices; returns its JSON object or {"error": ...} -- an optional extra must never be able
    to take the line down with it"""
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__), "--group-child", "--config", args.config, "--group-shards", str(shards)]
    if args.tune:
        cmd += ["--tune", args.tune]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE",
                                                             "GROUP_RANK", "ROLE_RANK", "ROLE_WORLD_SIZE", "TORCHELASTIC_RUN_ID")}
    t0 = time.perf_counter()
    try:
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        out = json.loads(lines[-1]) if (p.returncode == 0 and lines) else {"error": f"child exited with {p.returncode}: {p.stderr[-400:]}"}
    except subprocess.TimeoutExpired:
        out = {"error": f"child process exceeded its {timeout:.0f} s deadline and was stopped"}
    except Exception as exc:  # noqa: BLE001
        out = {"error": repr(exc)}
    out["child_seconds"] = round(time.perf_counter() - t0, 2)
    return out


def workload_name(config, N, C, b, cf):
    n_keys, arity, value_bytes = CONFIGS[config]
    return (f"{config}: 2^{n_keys.bit_length() - 1} keys, 32 B key / {value_bytes} B value, {arity}-wise XOR BFF; "
            f"encoded DB N={N} x C={C}, b={b}, {cf} fields/u32, packed D^T {4 * C * -(-N // cf) / 1e9:.3f} GB")


def respond_roofline(C, cf, b, shard_slots, layout, shard_words, passes_per_launch, launch_us, force_slice=False):
    """The respond kernel against the HBM roof, per LAUNCH (one launch = `passes_per_launch` independent passes over this GPU's shard).
    `achieved` / `frac` are a RATE: the bytes the launch really has to move with the packing that is resident (device image incl. zero
    padding + the query slice + the response; the PMC counters read 1.000 x this figure, `traffic`) over the launch time, against the 8 TB/s
    spec -- it cannot exceed 1.  SURVEY.md 8(d)'s figure -- the ALGORITHMIC bytes of the reference's packing (cf fields per u32) over the same
    time -- is kept beside it as `achieved_algorithmic_equiv` / `frac_algorithmic_equiv`: the planar image is 0.848 x the reference packing at
    b = 9, so that figure says how fast a kernel streaming the REFERENCE's bytes would have to be, not what moves.  Where the bytes of a
    launch do not come from HBM at all (working set of a pass inside the 256 MiB Infinity Cache, or passes walked in the interleaved
    order, which share database bytes on die) every `frac*` is null: there is no HBM rate to state (`achieved` still says what the kernel consumes)."""
    W_shard = -(-shard_slots // cf) if shard_slots > 0 else 0
    alg_q = 4 * C * W_shard + 4 * shard_slots + 4 * C  # SURVEY.md 8(d), this rank's shard
    moved_q = 4 * shard_words + 4 * shard_slots + 4 * C
    alg, moved = alg_q * passes_per_launch, moved_q * passes_per_launch
    rate = (lambda nbytes: nbytes / (launch_us * 1e-6) / 1e9 if launch_us > 0 else 0.0)
    packing = {0: "reference", 1: "dense64", 2: "planar"}[int(layout.packing)]
    mall = bool(moved_q <= 256 * (1 << 20))
    # order in which one launch walks its passes: "slice" = every pass is its own stream from HBM;
    # "interleaved" = concurrent passes share database bytes on die (chosen below ~1 GB per pass)
    order = "interleaved" if (passes_per_launch > 1 and 4 * shard_words <= (960 << 20) and not force_slice) else "slice"
    on_die = mall or order == "interleaved"
    return {
        "bound": "hbm",
        "kernel": "respond_planar_wide_kernel" if int(layout.packing) == 2 else "respond_kernel",
        "achieved": round(rate(moved), 1),
        "peak": HBM_PEAK_GBPS,
        "unit": "GB/s",
        "frac": None if on_die else round(rate(moved) / HBM_PEAK_GBPS, 4),
        "frac_is": ("bytes really moved (resident image + query slice + response) / launch time / 8 TB/s: a rate, <= 1" if not on_die else
                    "null: the launch's bytes are served on die (Infinity-Cache-sized pass or interleaved pass order), not an HBM rate"),
        "traffic": None,
        "launch_us": round(launch_us, 2),
        "passes_per_launch": passes_per_launch,
        "us_per_query": round(launch_us / passes_per_launch, 2),
        "packing": (f"planar (low byte + {b - 8} bit plane(s) per field = {b} bits, i8 MFMA operand order)" if int(layout.packing) == 2 else
                    f"{packing} ({layout.fields_per_word} fields per {'u64' if int(layout.packing) == 1 else 'u32'})"),
        "moved_bytes_per_launch": moved,
        "moved_GBps": round(rate(moved), 1),
        "frac_moved": None if on_die else round(rate(moved) / HBM_PEAK_GBPS, 4),  # = frac (kept for readers of earlier rounds' lines)
        # algorithmic bytes of the reference packing over the same launch time (NOT a rate of bytes that move)
        "bytes_per_launch": alg,
        "achieved_algorithmic_equiv": round(rate(alg), 1),
        "frac_algorithmic_equiv": None if on_die else round(rate(alg) / HBM_PEAK_GBPS, 4),
        "on_die": on_die,  # the launch's bytes are served from L2 / Infinity Cache: `achieved` is what the kernel consumes, no fraction of the HBM roof is stated
        "moved_over_algorithmic": round(moved / alg, 4) if alg else None,
        "traffic_over_algorithmic": None,
        "mall_resident": mall,
        "pass_order": order,
    }


class RespondLoop:
    """The timed loop of the bench, shared by the headline and by the sections that run OTHER configs the same way: a step = `qps_step`
    distinct queries of the pool answered by this rank's shard (one C call enqueues the step's passes) and, with several ranks, ONE
    integer all-reduce of the step's partial responses, overlapped with the next step's launches (two response buffers)."""

    def __init__(self, torch, dist, sharded, q_pool, qps_step, C, world, stream, enqueue="batch"):
        self.torch, self.dist, self.sharded, self.q_pool, self.qps_step, self.world, self.stream, self.enqueue = torch, dist, sharded, q_pool, qps_step, world, stream, enqueue
        self.pool = q_pool.shape[0]
        # two response buffers: with several ranks the all-reduce of step k overlaps the respond launches of step k+1
        self.r_bufs = [torch.zeros((qps_step, C), dtype=torch.int32, device="cuda") for _ in range(2)]
        self.pending = [None, None]
        self.step_counter = [0]

    def drain(self):
        for i in range(2):
            if self.pending[i] is not None:
                self.pending[i].wait()
                self.pending[i] = None

    def run_step(self, events=None, out=None):
        k = self.step_counter[0]
        qps_step, stream = self.qps_step, self.stream
        base = (k * qps_step) % self.pool
        self.step_counter[0] += 1
        buf = k % 2 if out is None else 0
        r = self.r_bufs[buf] if out is None else out
        if self.pending[buf] is not None:  # the collective that last used this buffer must be done before it is overwritten
            self.pending[buf].wait()
            self.pending[buf] = None
        if events:
            events[0].record(stream)
        if self.enqueue == "batch":
            self.sharded.respond_partial_device(self.q_pool[base:base + qps_step], r, batch=qps_step, stream=stream)
        else:
            for j in range(qps_step):
                self.sharded.respond_partial_device(self.q_pool[base + j], r[j], stream=stream)
        if events:
            events[1].record(stream)
        if self.world > 1:
            # int32 sum == u32 wrap-around sum; ONE collective for the step's queries, overlapped with the next step
            self.pending[buf] = self.dist.all_reduce(r, async_op=True)

    def timed_region(self, warmup, steps):
        """W untimed warm-up steps, then exactly K steps bracketed by a barrier + device synchronise on both sides; returns the wall time
        (MAX over ranks) and the summed HIP-event time of the respond launches on their stream"""
        torch, dist, world = self.torch, self.dist, self.world
        for _ in range(warmup):
            self.run_step()
        self.drain()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        # HIP events on the stream the respond kernels are launched on, bracketing the kernel launches of every timed step
        events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        # (torch creates the underlying HIP event at the first record(): the measuring apparatus is built BEFORE the timed region, not inside it;
        # measured: it makes no difference to the figure -- the first region of a process runs ~2 % below the next ones either way, the device
        # needs more than the contract's W = 5 warm-up steps to reach its steady clocks: `value_samples`)
        for a, b in events:
            a.record(self.stream), b.record(self.stream)
        torch.cuda.synchronize()
        t_begin = time.perf_counter()
        for k in range(steps):
            self.run_step(events[k])
        self.drain()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t_begin
        region_ms = sum(a.elapsed_time(b) for a, b in events)
        if world > 1:
            t = torch.tensor([wall], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wall = float(t.item())
        return wall, region_ms


def config_section(cp, torch, dist, device, name, rank, world, stream, steps=5, warmup=2, per_gpu=32, single_ref=False, tune=""):
    """ANOTHER config of BASELINE.json run the headline's way inside this process: this rank's shard of the synthetic encoded database
    generated in HBM and packed, `per_gpu` queries per GPU and step as that many independent passes in SLICE order (every pass its own
    stream of the shard from HBM, nt loads -- whatever the product would dispatch at this shard size, which is timed beside it where it
    differs), W warm-up steps and K timed steps between barriers, one all-reduce per step with several ranks; then the exact-sum check
    (unit / dense / all-ones queries against 64-bit sums of the regenerated rows; `multirank_bit_exact` with several ranks) and, on
    request, rank 0's own run of the WHOLE database on its one GPU (like for like).  N = 1: `other_configs` of the default line (the
    reference's own grid has 2^20 keys x 4-wise, integrations/benches/online_phase.rs:40-57).  N > 1: `baseline_multi_gpu_configs` --
    BASELINE.json configs[3] (2^22 keys) and configs[4] (8 kB values) are DEFINED as 8-GPU configs."""
    from chalametpir_amd.distributed import ShardedServer, shard_range

    t_begin = time.perf_counter()
    n_keys, arity, value_bytes = CONFIGS[name]
    b = cp.find_encoded_db_matrix_element_bit_length(n_keys)
    _, _, N = cp.filter_shape(arity, n_keys)
    C = cp.encoded_num_cols(value_bytes, b)
    cf = 2 if b >= 11 else (3 if b >= 9 else 4)
    mask = (1 << b) - 1
    layout = cp.dtc_layout_for(N, C, b)
    lo, hi = shard_range(N, layout, rank, world)
    D = torch.empty(((hi - lo), C), dtype=torch.int32, device="cuda")
    if hi > lo:
        device.synth_fill(D, (hi - lo) * C, SEED_D, index0=lo * C, mask=mask, stream=stream)
    sharded = ShardedServer.from_device_matrix(D, lo, hi, C, b, N, device, stream=stream)
    torch.cuda.synchronize()
    del D
    torch.cuda.empty_cache()
    qps_step = per_gpu * world
    pool = qps_step if world > 1 else 2 * qps_step  # distinct queries cycled through
    q_pool = torch.empty((pool, N), dtype=torch.int32, device="cuda")
    for i in range(pool):
        device.synth_fill(q_pool, N, SEED_Q + i, offset_words=i * N, stream=stream)
    torch.cuda.synchronize()
    cp.tuning_set("respond.batch_fusion", 0)
    loop = RespondLoop(torch, dist, sharded, q_pool, qps_step, C, world, stream)
    shard_words = int(sharded.local.layout.total_words) if sharded.local is not None else 0
    n_queries = steps * qps_step
    user_order = next((int(kv.split("=")[1]) for kv in tune.split(",") if kv.startswith("respond.interleave_passes=")), -1)
    out = {"workload": workload_name(name, N, C, b, cf), "n_gpus": world, "steps": steps, "warmup": warmup, "queries_per_step": qps_step,
           "shard_slots_rank0": hi - lo if rank == 0 else None, "resident_bytes_per_gpu_rank0": 4 * shard_words}
    try:
        cp.tuning_set("respond.interleave_passes", 0)
        wall, region_ms = loop.timed_region(warmup, steps)
        roof = respond_roofline(C, cf, b, hi - lo, layout, shard_words, qps_step, region_ms * 1e3 / steps, force_slice=True)
        out.update({
            "value": round(n_queries / wall, 2), "unit": "queries/s", "pass_order": "slice", "ms_per_step": round(wall / steps * 1e3, 4),
            "launch_us": roof["launch_us"], "us_per_query_per_gpu": roof["us_per_query"], "passes_per_launch": qps_step,
            "frac": roof["frac"], "frac_moved": roof["frac_moved"], "frac_algorithmic_equiv": roof["frac_algorithmic_equiv"],
            "moved_GBps_per_gpu": roof["moved_GBps"], "moved_bytes_per_launch": roof["moved_bytes_per_launch"], "mall_resident": roof["mall_resident"],
            "algorithmic_bytes_per_query": 4 * C * -(-N // cf) + 4 * N + 4 * C,
        })
        as_dispatched = respond_roofline(C, cf, b, hi - lo, layout, shard_words, qps_step, 1.0)["pass_order"]
        if as_dispatched != "slice":  # what the product dispatches at this shard size (passes interleaved: database bytes shared on die)
            cp.tuning_set("respond.interleave_passes", user_order)
            wall_d, _ = loop.timed_region(1, steps)
            out["value_as_dispatched"] = round(n_queries / wall_d, 2)
            out["as_dispatched_pass_order"] = as_dispatched
    finally:
        cp.tuning_set("respond.interleave_passes", user_order)
    loop.drain()
    del loop
    try:
        chk = multirank_check(torch, dist, device, sharded, N, C, b, mask, lo, hi, rank, world, layout, stream)
    except Exception as exc:  # noqa: BLE001 -- a check that could not run is reported as such
        log(f"rank {rank}: {name}: exact-sum check failed to run: {exc!r}")
        chk = {"multirank_bit_exact": None, "error": repr(exc)}
    if world > 1:
        out["multirank_bit_exact"], out["ranks_seen"] = chk.get("multirank_bit_exact"), chk.get("ranks_seen")
    else:
        out["responses_bit_exact_vs_64bit_sums"] = chk.get("multirank_bit_exact")
    out["check"] = {k: chk.get(k) for k in ("unit_queries", "unit_queries_ok", "dense_and_all_ones_queries", "dense_and_all_ones_ok",
                                            "same_response_on_every_rank", "shard_slots", "seconds", "error") if k in chk}
    del sharded
    torch.cuda.empty_cache()
    if single_ref and world > 1:
        # rank 0 alone: the WHOLE database of this config on its one GPU, the same loop (the other ranks wait at the barrier)
        if rank == 0:
            try:
                ref = single_gpu_reference(cp, torch, device, ShardedServer, steps, tune, N, C, b, mask, q_pool, stream)
                out["single_gpu_reference"] = ref
                out["scaling_like_for_like"] = round(out["value"] / ref["queries_per_sec"], 3)
            except Exception as exc:  # noqa: BLE001
                log(f"{name}: single-GPU reference failed: {exc!r}")
                out["single_gpu_reference"] = {"error": repr(exc)}
        dist.barrier()
    del q_pool
    torch.cuda.empty_cache()
    out["seconds_spent"] = round(time.perf_counter() - t_begin, 2)
    return out


def launch_ranks(n: int) -> int:
    """one rank per GPU through torch.distributed.run on 127.0.0.1 (the container hostname may not resolve); the child's stdout and
    stderr are this process' own, so rank 0's JSON line is the only line on stdout; returns the child's exit code"""
    import socket
    import subprocess

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"bench.py: --gpus {n} without a launcher: starting {n} ranks: {' '.join(cmd)}")
    return subprocess.run(cmd, env=env).returncode


def single_gpu_reference(cp, torch, device, ShardedServer, steps, tune, N, C, b, mask, q_pool, stream):
    """Rank 0 alone, after the timed regions: the WHOLE synthetic database on this one GPU, 32 queries per step as 32 independent passes in
    slice order (every query its own stream from HBM) -- the N = 1 headline's loop, measured in this process, so that an N > 1 line can
    state its scaling against it without a second run (`scaling_like_for_like`)."""
    t0 = time.perf_counter()
    D = torch.empty((N, C), dtype=torch.int32, device="cuda")
    device.synth_fill(D, N * C, SEED_D, mask=mask, stream=stream)
    whole = ShardedServer.from_device_matrix(D, 0, N, C, b, N, device, stream=stream)
    torch.cuda.synchronize()
    del D
    torch.cuda.empty_cache()
    per_step = min(32, q_pool.shape[0])
    pool = q_pool.shape[0] // per_step * per_step
    r = torch.zeros((per_step, C), dtype=torch.int32, device="cuda")
    cp.tuning_set("respond.batch_fusion", 0)
    cp.tuning_set("respond.interleave_passes", 0)

    def step(k):
        base = (k * per_step) % pool
        whole.respond_partial_device(q_pool[base:base + per_step], r, batch=per_step, stream=stream)

    try:
        for k in range(3):
            step(k)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for k in range(steps):
            step(k)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t1
    finally:
        user_order = next((int(kv.split("=")[1]) for kv in tune.split(",") if kv.startswith("respond.interleave_passes=")), -1)
        cp.tuning_set("respond.interleave_passes", user_order)
    qps = steps * per_step / wall
    del whole, r
    torch.cuda.empty_cache()
    return {"queries_per_sec": round(qps, 2), "us_per_query": round(1e6 / qps, 2), "steps": steps, "queries_per_step": per_step,
            "pass_order": "slice", "seconds_spent": round(time.perf_counter() - t0, 2),
            "note": "the whole database on rank 0's GPU alone, same generator, same loop as the N = 1 headline, measured in this process after "
                    "the timed regions"}


def kernel_source_sha256() -> str:
    """fingerprint of the respond kernels' sources: ties a committed counter pass to the code it measured"""
    import hashlib

    h = hashlib.sha256()
    for name in ("respond_planar.hip", "respond.hip"):
        with open(os.path.join(ROOT, "chalametpir_amd", "csrc", name), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def committed_traffic(config: str, algorithmic_bytes_per_pass: int, packing: str):
    """the record of profiles/respond_traffic.json for this workload, or None"""
    try:
        with open(os.path.join(ROOT, "profiles", "respond_traffic.json")) as fh:
            doc = json.load(fh)
    except (OSError, ValueError):
        return None
    for rec in doc.get("records", []):
        if (rec.get("config") == config and int(rec.get("algorithmic_bytes_per_pass", 0)) == algorithmic_bytes_per_pass
                and rec.get("packing") == packing):
            return rec
    return None


def live_traffic(args, passes_per_launch: int):
    """HBM bytes per LAUNCH of the headline respond kernel, measured now: this script's timed loop alone (`--headline-only`) is run twice as
    a child process under `rocprofv3 --kernel-trace --pmc <counter>` (FETCH_SIZE, then WRITE_SIZE: separate passes, kernel trace only), and
    the counters of the dominant respond kernel are averaged over its dispatches.  gfx950: FETCH_SIZE counts 64 B per 128 B request of a
    16 B/lane coalesced stream -> x2 (MI355X_MICROARCH.md, HBM section); both counters are in units of 1024 B.  None if the profiler is not
    there or a pass fails -- an optional extra must never take the headline down."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None
    # not from inside a profiled run (the profiler's preloaded tool library and its environment would be inherited by the children)
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", "") or os.environ.get("HSA_TOOLS_LIB"):
        log("live traffic: this process is being profiled itself; quoting the committed counter pass instead")
        return None
    child = [sys.executable, os.path.abspath(__file__), "--headline-only", "--no-setup", "--no-setup-kv", "--no-cpu-baseline", "--no-host-path", "--no-read-ceiling",
             "--no-live-traffic", "--config", args.config, "--steps", "3", "--warmup", "1", "--queries-per-step", str(args.queries_per_step),
             "--query-pool", str(args.query_pool), "--enqueue", args.enqueue]
    if args.tune:
        child += ["--tune", args.tune]
    means = {}
    tmp = tempfile.mkdtemp(prefix="cpir_traffic_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [prof, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "t", "--"] + child
            # its own session: on expiry the whole process GROUP goes (the profiler AND the profiled python under it -- killing only
            # rocprofv3 would leave the child running on the GPU beside the sections timed next)
            proc = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                    start_new_session=True)
            try:
                _, err = proc.communicate(timeout=150)
            except subprocess.TimeoutExpired:
                import signal

                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except OSError:
                    pass
                proc.communicate()
                log(f"live traffic: the {counter} pass exceeded 150 s and was stopped (process group {proc.pid}); quoting the committed counter pass")
                return None
            if proc.returncode != 0:
                log(f"live traffic: rocprofv3 --pmc {counter} exited with {proc.returncode}: {err[-300:]}; quoting the committed counter pass")
                return None
            per_kernel = {}
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        name = row.get("Kernel_Name") or ""
                        if (row.get("Counter_Name") or "") == counter and ("respond_planar" in name or "respond_kernel" in name):
                            per_kernel.setdefault(name, []).append(float(row.get("Counter_Value") or 0))
            if not per_kernel:
                log(f"live traffic: no respond kernel in the {counter} pass")
                return None
            name = max(per_kernel, key=lambda k: sum(per_kernel[k]))  # the instantiation that carries the timed loop
            means[counter] = (name, sum(per_kernel[name]) / len(per_kernel[name]), len(per_kernel[name]))
    except Exception as exc:  # noqa: BLE001
        log(f"live traffic: {exc!r}")
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    fetch, write = means["FETCH_SIZE"], means["WRITE_SIZE"]
    import re

    m = re.search(r"respond_\w+(<[^>]*>)?", fetch[0])
    short = m.group(0) if m else fetch[0][:60]
    return {
        "bytes_per_launch": (2 * fetch[1] + write[1]) * 1024,
        "source": (f"measured in this run: the timed loop alone in two child processes under rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE "
                   f"(separate passes), mean over {fetch[2]} / {write[2]} dispatches of {short}; FETCH_SIZE x2 (gfx950: 64 B "
                   f"counted per 128 B request of a 16 B/lane stream) + WRITE_SIZE, units of 1024 B; one launch = {passes_per_launch} passes"),
    }


def mix_ceiling(read_bytes: int):
    """what a plain mixed stream (7 reads : 2 writes, everything coalesced) moves on this device right now: the yardstick of the pack pass"""
    import subprocess

    tool = os.path.join(ROOT, "chalametpir_amd", "lib", "hbm_read_ceiling")
    if not os.path.exists(tool):
        return None
    try:
        p = subprocess.run([tool, "--mix", str(int(read_bytes))], capture_output=True, text=True, timeout=90)
        return json.loads(p.stdout.strip().splitlines()[-1]) if p.returncode == 0 else None
    except Exception:  # noqa: BLE001 -- an optional extra must never take the headline down
        return None


def read_ceiling(nbytes: int):
    """what a bare read-only kernel gets from this device's HBM right now (chalametpir_amd/lib/hbm_read_ceiling, a child process)"""
    import subprocess

    tool = os.path.join(ROOT, "chalametpir_amd", "lib", "hbm_read_ceiling")
    if not os.path.exists(tool):
        return None
    try:
        p = subprocess.run([tool, "--json", str(int(nbytes))], capture_output=True, text=True, timeout=60)
        return json.loads(p.stdout.strip().splitlines()[-1]) if p.returncode == 0 else None
    except Exception:  # noqa: BLE001 -- an optional extra must never take the headline down
        return None


def verify(run_step, drain, step_counter, r_step, qps_step, pool, N, C, b, mask, rank, torch):
    """One more step; rank 0 rebuilds the FULL synthetic DB and the step's queries on the host with the oracle's copy of the
    generator, and checks the (all-reduced) responses bit for bit."""
    base = (step_counter[0] * qps_step) % pool
    run_step(out=r_step)
    drain()
    torch.cuda.synchronize()
    if rank != 0:
        return None
    from oracle import oracle as orc  # checker only

    D = orc.synth_fill_u32(N * C, SEED_D, 0, mask).reshape(N, C)
    dtc = orc.row_wise_compress(orc.transpose(D), b)
    got = r_step.cpu().numpy().view(np.uint32)
    ok = True
    for j in range(qps_step):
        q = orc.synth_fill_u32(N, SEED_Q + base + j)
        ok &= bool(np.array_equal(got[j], orc.row_vector_x_compressed_transposed_matrix(q, dtc, N, b)[0]))
    return ok


def synth_u32_np(indices, seed: int, mask: int = 0xFFFFFFFF) -> np.ndarray:
    """numpy statement of the counter-based generator (csrc/synth.hip: hi32 of the splitmix64 finaliser of (seed, index)): lets rank 0
    rebuild any row of the synthetic database without the device and without the oracle"""
    idx = np.asarray(indices, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) * np.uint64(0xD1342543DE82EF95) + (idx + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return ((z >> np.uint64(32)).astype(np.uint32)) & np.uint32(mask)


def multirank_check(torch, dist, device, sharded, N, C, b, mask, lo, hi, rank, world, layout, stream):
    """Default-on proof that the N-sharded respond + the integer all-reduce is bit-exact, independent of the database's size (runs at
    every config, after the timed region; nothing here touches the oracle):
      * UNIT queries k * e_n, n = first and last slot of every rank's shard (so every rank's kernel and every seam between shards is
        hit), k arbitrary u32 incl. values >= 2^31: the expected response k * D[n][:] mod 2^32 comes from the numpy statement of the
        counter-based generator alone, on rank 0;
      * two DENSE queries and the ALL-ONES query q = 0xFFFFFFFF (every product and every partial sum wraps): every rank sums its own
        shard exactly in 64 bits with torch from the regenerated, UNPACKED rows of D, the int64 partials are all-reduced and truncated,
        and compared with what the product's kernels + the int32-view all-reduce gave -- so the collective's SUM is shown to wrap like
        u32 addition on this backend and hardware;
      * both dispatch modes of the batch entry point (independent passes, fused passes);
      * every rank must hold the SAME reduced responses (min / max of a checksum over ranks).
    With world = 1 (the `other_configs` of the default line) the same queries check the single GPU's kernels against the same exact sums."""
    from chalametpir_amd.distributed import allreduce_u32_, shard_range
    import chalametpir_amd as cp

    t0 = time.perf_counter()
    bounds = [shard_range(N, layout, r, world) for r in range(world)]
    seen = torch.zeros(world, dtype=torch.int64, device="cuda")
    seen[rank] = 1 + (hi - lo)
    if world > 1:
        dist.all_reduce(seen)
    seen = seen.cpu().tolist()
    ranks_seen = [r for r in range(world) if seen[r] > 0]
    slots_ok = all(seen[r] == 1 + (bounds[r][1] - bounds[r][0]) for r in range(world))

    slots = sorted({n for a, z in bounds if z > a for n in (a, z - 1)} | {0, N - 1})
    ks = [((0x9E3779B1 * (i + 1)) ^ (0x80000000 if i & 1 else 0) | 1) & 0xFFFFFFFF for i in range(len(slots))]
    n_unit, n_sum = len(slots), 3
    nq = n_unit + n_sum
    q = torch.zeros((nq, N), dtype=torch.int32, device="cuda")
    for i, (n, k) in enumerate(zip(slots, ks)):
        q[i, n] = k - (1 << 32) if k >= (1 << 31) else k
    for j in range(2):
        device.synth_fill(q, N, 0xC0DE00 + j, offset_words=(n_unit + j) * N, stream=stream)
    q[nq - 1].fill_(-1)  # 0xFFFFFFFF everywhere
    torch.cuda.synchronize()

    got = {}
    for fusion in (0, 1):
        cp.tuning_set("respond.batch_fusion", fusion)
        r = torch.zeros((nq, C), dtype=torch.int32, device="cuda")
        sharded.respond_partial_device(q, r, batch=nq, stream=stream)
        if world > 1:
            allreduce_u32_(r)  # the product's exchange step: int32-view SUM over the process group
        torch.cuda.synchronize()
        got[fusion] = r.to(torch.int64) & 0xFFFFFFFF
    cp.tuning_set("respond.batch_fusion", 0)

    # exact 64-bit sums over this rank's rows of D, regenerated unpacked (int64 wrap-around is harmless: only the low 32 bits are kept)
    part = torch.zeros((n_sum, C), dtype=torch.int64, device="cuda")
    rows_per = max(1, (64 << 20) // (8 * C))
    for a in range(0, hi - lo, rows_per):
        z = min(hi - lo, a + rows_per)
        Dc = torch.empty((z - a, C), dtype=torch.int32, device="cuda")
        device.synth_fill(Dc, (z - a) * C, SEED_D, index0=(lo + a) * C, mask=mask, stream=stream)
        Dc = Dc.to(torch.int64)
        for j in range(n_sum):
            qc = q[n_unit + j, lo + a:lo + z].to(torch.int64) & 0xFFFFFFFF
            part[j] += (qc[:, None] * Dc).sum(dim=0)
        del Dc
    if world > 1:
        dist.all_reduce(part)  # int64 sum
    want_sum = part & 0xFFFFFFFF

    same_on_all_ranks = True
    for fusion in (0, 1):
        ck = torch.stack([got[fusion].sum(), -got[fusion].sum()])
        if world > 1:
            dist.all_reduce(ck, op=dist.ReduceOp.MAX)  # max(x) == -max(-x)  <=>  every rank holds the same checksum
        same_on_all_ranks &= bool((ck[0] + ck[1]).item() == 0)

    sums_ok = all(bool(torch.equal(got[f][n_unit:], want_sum)) for f in (0, 1))
    units_ok = True
    if rank == 0:
        for f in (0, 1):
            g = got[f][:n_unit].cpu().numpy().astype(np.uint64)
            for i, (n, k) in enumerate(zip(slots, ks)):
                row = synth_u32_np(np.arange(n * C, (n + 1) * C, dtype=np.uint64), SEED_D, mask).astype(np.uint64)
                units_ok &= bool(np.array_equal(g[i], (row * np.uint64(k)) & np.uint64(0xFFFFFFFF)))
    flags = torch.tensor([int(sums_ok), int(units_ok), int(same_on_all_ranks), int(slots_ok)], dtype=torch.int64, device="cuda")
    if world > 1:
        dist.all_reduce(flags, op=dist.ReduceOp.MIN)
    sums_ok, units_ok, same_on_all_ranks, slots_ok = (bool(v) for v in flags.cpu().tolist())
    del q, part
    torch.cuda.empty_cache()
    return {
        "multirank_bit_exact": bool(sums_ok and units_ok and same_on_all_ranks and slots_ok and len(ranks_seen) == world),
        "ranks_seen": ranks_seen,
        "shard_slots": [z - a for a, z in bounds],
        "unit_queries": n_unit,
        "unit_queries_ok": units_ok,
        "dense_and_all_ones_queries": n_sum,
        "dense_and_all_ones_ok": sums_ok,
        "wraparound_case": "q = 0xFFFFFFFF in every slot: every product and every partial sum exceeds 2^32",
        "same_response_on_every_rank": same_on_all_ranks,
        "dispatch_modes": ["independent passes", "fused passes"],
        "seconds": round(time.perf_counter() - t0, 3),
        "expected_from": "unit queries: numpy statement of the counter-based generator on rank 0; dense / all-ones: exact int64 sums over each "
                         "rank's regenerated unpacked rows of D (torch), all-reduced as int64, truncated to 32 bits",
    }


def host_path_timing(server, q_pool, N, torch, full=True):
    """Server::respond as the drop-in sees it (cpir_server_respond: host query in, host response out, PCIe both ways):
    latency of one caller, and throughput with 8 and 16 concurrent callers on the one handle (the reference serves an
    Arc<Server> from many tokio tasks), from pageable and from page-locked query buffers; next to it the host link itself
    (one large page-locked copy each way).  Never the headline `value`: the headline has q resident in HBM."""
    import threading

    import chalametpir_amd as cp

    # the link: 256 MiB page-locked <-> device, HIP events on the copy stream
    nbig = (64 << 20) if full else (8 << 20)
    hp = torch.empty(nbig, dtype=torch.int32).pin_memory()
    dv = torch.empty(nbig, dtype=torch.int32, device="cuda")
    cs = torch.cuda.Stream()
    rates = {}
    for name, src, dst in (("h2d", hp, dv), ("d2h", dv, hp)):
        with torch.cuda.stream(cs):
            dst.copy_(src, non_blocking=True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(cs)
            for _ in range(4):
                dst.copy_(src, non_blocking=True)
            e1.record(cs)
        cs.synchronize()
        rates[name] = 4 * nbig * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del hp, dv

    qs = [q_pool[i].cpu().numpy().view(np.uint32) for i in range(min(16, q_pool.shape[0]))]
    for q in qs[:4]:
        server.respond_array(q)
    n1 = 64

    def closed_loop(get):  # mean of n1 calls back to back (the figure a closed-loop caller sees) and their median (one stall in 64 calls
        ts = []            # -- a helper thread losing its core for a few ms -- moves the mean by 50 us and the median not at all)
        t_start = time.perf_counter()
        for i in range(n1):
            t_a = time.perf_counter()
            server.respond_array(get(i))
            ts.append(time.perf_counter() - t_a)
        return (time.perf_counter() - t_start) / n1, float(np.median(ts))

    lat, lat_med = closed_loop(lambda i: qs[i % len(qs)])
    # the same single caller with its query in page-locked memory (cpir_host_alloc): DMA straight from the caller's buffer
    pins = [cp.PinnedArray(N) for _ in range(16)]
    for i, p in enumerate(pins):
        p.array[:] = qs[i % len(qs)]
    for _ in range(3):
        server.respond_array(pins[0].array)
    lat_pinned, lat_pinned_med = closed_loop(lambda i: pins[0].array)
    # the same with the in-place read switched off (upload first, as concurrent callers do): what the zero-copy path saves
    cp.tuning_set("respond.host_zero_copy", 0)
    try:
        for _ in range(3):
            server.respond_array(pins[0].array)
        t0 = time.perf_counter()
        for i in range(n1):
            server.respond_array(pins[0].array)
        lat_pinned_upload = (time.perf_counter() - t0) / n1
    finally:
        cp.tuning_set("respond.host_zero_copy", 1)

    def throughput(threads, per, pinned):
        def work(k):
            for i in range(per):
                server.respond_array(pins[k].array if pinned else qs[(k + i) % len(qs)])

        ts = [threading.Thread(target=work, args=(k,)) for k in range(threads)]
        t0 = time.perf_counter()
        [t.start() for t in ts]
        [t.join() for t in ts]
        return threads * per / (time.perf_counter() - t0)

    throughput(8, 4, False)  # warm-up (first use of every arena)
    if not full:  # the short form (the real-database section): one caller both ways, eight callers
        out = {
            "one_caller_us_per_query": round(lat * 1e6, 1),
            "one_caller_pinned_query_us_per_query": round(lat_pinned * 1e6, 1),
            "one_caller_median_us": round(lat_med * 1e6, 1),
            "one_caller_pinned_query_median_us": round(lat_pinned_med * 1e6, 1),
            "one_caller_pinned_query_upload_first_us_per_query": round(lat_pinned_upload * 1e6, 1),
            "eight_callers_queries_per_sec": round(throughput(8, 48, False), 1),
            "eight_callers_pinned_queries_per_sec": round(throughput(8, 48, True), 1),
            "host_gather": cp.host_gather_variant(),
        }
        for p in pins:
            p.close()
        return out
    out = {
        "one_caller_us_per_query": round(lat * 1e6, 1),
        "one_caller_queries_per_sec": round(1.0 / lat, 1),
        "one_caller_pinned_query_us_per_query": round(lat_pinned * 1e6, 1),
        "one_caller_median_us": round(lat_med * 1e6, 1),
        "one_caller_pinned_query_median_us": round(lat_pinned_med * 1e6, 1),
        "one_caller_pinned_query_upload_first_us_per_query": round(lat_pinned_upload * 1e6, 1),
        "four_callers_queries_per_sec": round(throughput(4, 64, False), 1),
        "four_callers_pinned_queries_per_sec": round(throughput(4, 64, True), 1),
        "eight_callers_queries_per_sec": round(throughput(8, 48, False), 1),
        "eight_callers_pinned_queries_per_sec": round(throughput(8, 48, True), 1),
        "sixteen_callers_queries_per_sec": round(throughput(16, 24, False), 1),
        "sixteen_callers_pinned_queries_per_sec": round(throughput(16, 24, True), 1),
        "query_bytes": 4 * N,
        "h2d_GBps": round(rates["h2d"], 1),
        "d2h_GBps": round(rates["d2h"], 1),
        "link_bound_queries_per_sec": round(rates["h2d"] * 1e9 / (4 * N), 1),
        "note": "cpir_server_respond on host buffers.  A lone caller is served without an upload: the step-major kernel reads the query in "
                "place over the host link (from the caller's page-locked buffer, or from the server's pinned block WHILE the caller's pageable query is "
                "being copied into it: one launch in front of the copy, the kernel polling the copy's progress) + D2H. "
                "Two to four concurrent callers (respond.inplace_seats): ONE pass reads their queries in place -- page-locked ones from the callers' "
                "buffers, pageable ones from the pinned block while their callers copy them in (every seat's progress polled). More "
                "concurrent callers: pinned staging (skipped for page-locked queries; on a server with a slot map the query is compacted while it "
                "is staged) + H2D + batched respond + D2H, coalesced into arenas of up to 8 seats, uploads on two streams taken in turn, kernels "
                "back to back on another; link_bound = h2d_GBps / query_bytes",
        "concurrent_callers_are": "PYTHON threads: each call holds the interpreter lock for 50-100 us around the library call, which caps this "
                                  "harness near 7-9 k calls/s whatever the library does (16 callers read LOWER than 8 for that reason alone). The "
                                  "library's own concurrency figures are respond_host_path_native / _native_compacted (plain C threads, same entry point)",
    }
    for p in pins:
        p.close()
    return out


def group_child(cp, torch, device, args):
    """--group-child: build the synthetic database and queries again in this process, a single-device server to compare with, and
    time the group path; prints one JSON object"""
    n_keys, arity, value_bytes = CONFIGS[args.config]
    b = cp.find_encoded_db_matrix_element_bit_length(n_keys)
    _, _, N = cp.filter_shape(arity, n_keys)
    C = cp.encoded_num_cols(value_bytes, b)
    mask = (1 << b) - 1
    stream = torch.cuda.current_stream()
    D_dev = torch.empty((N, C), dtype=torch.int32, device="cuda")
    device.synth_fill(D_dev, N * C, SEED_D, mask=mask, stream=stream)
    single = cp.Server.from_device_matrix(D_dev, N, C, b, device=device, stream=stream)
    torch.cuda.synchronize()
    del D_dev
    torch.cuda.empty_cache()
    q_pool = torch.empty((16, N), dtype=torch.int32, device="cuda")
    for i in range(16):
        device.synth_fill(q_pool, N, SEED_Q + i, offset_words=i * N, stream=stream)
    torch.cuda.synchronize()
    n_vis = torch.cuda.device_count()
    out = group_host_path_timing(cp, torch, device, single, q_pool, N, C, b, mask, args.group_shards, n_vis, stream)
    single.close()
    del single, q_pool
    torch.cuda.empty_cache()
    if n_keys * (32 + value_bytes) <= (2 << 30):
        # ... and the group handle exactly as the drop-in builds it: from the KEY-VALUE database (cpir_server_setup_kv_multi)
        try:
            out["from_kv_database"] = group_kv_timing(cp, device, n_keys, arity, value_bytes, args.group_shards, n_vis)
        except Exception as exc:  # noqa: BLE001
            out["from_kv_database"] = {"error": repr(exc)}
    print(json.dumps(out), flush=True)
    return 0


def group_kv_timing(cp, device0, n_keys, arity, value_bytes, shards, n_vis):
    """`Server::setup::<ARITY>(seed, kv database)` over SEVERAL devices behind one handle -- cpir_server_setup_kv_multi, the call
    rust/server_hip.rs makes when CHALAMET_HIP_DEVICES names more than one -- next to the same setup on one device with the same filter
    seeds: the same filter parameters, the same hint, the same responses; then one host caller's latency and eight callers' throughput on
    the group (every shard keeps only its slots that hold something: the real database's slot maps, shard by shard)."""
    import threading

    keys, key_off, values, val_off = synthetic_kv_database(n_keys, value_bytes)
    seeds = np.random.default_rng(0xF117).integers(0, 256, size=32 * 100, dtype=np.uint8).tobytes()
    devs = [device0 if (i % n_vis) == device0.ordinal else cp.Device(i % n_vis) for i in range(shards)]
    t0 = time.perf_counter()
    grp, hint_g, filt_g = cp.Server.setup_flat(SEED_MU, keys, key_off, values, val_off, arity, devices=devs, filter_seed_material=seeds)
    wall_g = time.perf_counter() - t0
    t0 = time.perf_counter()
    one, hint_1, filt_1 = cp.Server.setup_flat(SEED_MU, keys, key_off, values, val_off, arity, device=device0, filter_seed_material=seeds)
    wall_1 = time.perf_counter() - t0
    del keys, values
    N = grp.decompressed_num_cols
    rng = np.random.default_rng(0x9E)
    qs = [rng.integers(0, 1 << 32, size=N, dtype=np.uint64).astype(np.uint32) for _ in range(8)]
    same = all(bool(np.array_equal(grp.respond_array(q), one.respond_array(q))) for q in qs[:3])
    for q in qs[:4]:
        grp.respond_array(q)
    n1 = 32
    t0 = time.perf_counter()
    for i in range(n1):
        grp.respond_array(qs[i % len(qs)])
    lat = (time.perf_counter() - t0) / n1
    threads, per = 8, 16

    def work(k):
        for i in range(per):
            grp.respond_array(qs[(k + i) % len(qs)])

    ts = [threading.Thread(target=work, args=(k,)) for k in range(threads)]
    t0 = time.perf_counter()
    [t.start() for t in ts]
    [t.join() for t in ts]
    thr = threads * per / (time.perf_counter() - t0)
    served, of = grp.slots_served()
    out = {
        "entry_point": "cpir_server_setup_kv_multi",
        "shards": grp.group_shards(),
        "server_setup_kv_multi_wall_sec": round(wall_g, 3),
        "server_setup_kv_single_device_wall_sec": round(wall_1, 3),
        "same_filter_params_as_single_device": filt_g == filt_1,
        "same_hint_as_single_device": hint_g == hint_1,
        "responses_equal_single_device": same,
        "slots_served": served, "slots_of": of,
        "one_caller_us_per_query": round(lat * 1e6, 1),
        "eight_callers_queries_per_sec": round(thr, 1),
        "note": "the multi-GPU path of the Rust drop-in, end to end on host bytes: Server::setup from the key-value database split over the listed "
                "devices, Server::respond scattering each query's slots over every device's own host link, partial responses summed on the host",
    }
    grp.close()
    one.close()
    return out


def group_host_path_timing(cp, torch, device0, single, q_pool, N, C, b, mask, shards, n_vis, stream):
    """Server::respond on host buffers through a GROUP handle: the database split over the devices of THIS process, every query
    scattered (device g receives only its slots of q over its own host link), partial responses summed on the host.  Checked
    against the single-device server on the same database first."""
    import threading

    D_dev = torch.empty((N, C), dtype=torch.int32, device="cuda")
    device0.synth_fill(D_dev, N * C, SEED_D, mask=mask, stream=stream)
    torch.cuda.synchronize()
    D_host = D_dev.cpu().numpy().view(np.uint32)
    del D_dev
    torch.cuda.empty_cache()
    devs = [device0 if (i % n_vis) == device0.ordinal else cp.Device(i % n_vis) for i in range(shards)]
    t0 = time.perf_counter()
    grp, hint = cp.Server.setup_from_matrix(SEED_MU, D_host, b, devices=devs)
    setup_wall = time.perf_counter() - t0
    del D_host
    qs = [q_pool[i].cpu().numpy().view(np.uint32) for i in range(min(16, q_pool.shape[0]))]
    # same answers as the single-device server the headline ran on (same synthetic database)?
    same = all(bool(np.array_equal(grp.respond_array(qs[i]), single.respond_array(qs[i]))) for i in range(3))
    n1 = 48
    for q in qs[:4]:
        grp.respond_array(q)
    t0 = time.perf_counter()
    for i in range(n1):
        grp.respond_array(qs[i % len(qs)])
    lat = (time.perf_counter() - t0) / n1
    threads, per = 8, 24

    def work(k):
        for i in range(per):
            grp.respond_array(qs[(k + i) % len(qs)])

    ts = [threading.Thread(target=work, args=(k,)) for k in range(threads)]
    t0 = time.perf_counter()
    [t.start() for t in ts]
    [t.join() for t in ts]
    thr = threads * per / (time.perf_counter() - t0)
    # the same handle asked on DEVICE pointers: the shards pull their slots of the queries from the root device over the peer link and push
    # their partial responses into its table, a kernel there adds them up (no host, no collective library)
    nb = min(16, q_pool.shape[0])
    rdev = torch.empty((nb, C), dtype=torch.int32, device="cuda")
    cp.tuning_set("respond.batch_fusion", 0)
    for _ in range(2):
        grp.respond_batch_device(q_pool[:nb], nb, rdev, stream=stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(8):
        grp.respond_batch_device(q_pool[:nb], nb, rdev, stream=stream)
    e1.record(stream)
    torch.cuda.synchronize()
    dev_us = e0.elapsed_time(e1) * 1e3 / (8 * nb)
    rsingle = torch.empty((nb, C), dtype=torch.int32, device="cuda")
    single.respond_batch_device(q_pool[:nb], nb, rsingle, stream=stream)
    torch.cuda.synchronize()
    dev_same = bool(torch.equal(rdev, rsingle))
    cp.tuning_set("respond.batch_fusion", 1)
    out = {
        "device_queries_us_per_query": round(dev_us, 2),
        "device_queries_equal_single_device": dev_same,
        "shards": grp.group_shards(),
        "visible_devices": n_vis,
        "one_caller_us_per_query": round(lat * 1e6, 1),
        "one_caller_queries_per_sec": round(1.0 / lat, 1),
        "eight_callers_queries_per_sec": round(thr, 1),
        "setup_wall_sec": round(setup_wall, 3),
        "hint_checksum": int(hint.sum(dtype=np.uint64) & 0xFFFFFFFFFFFFFFFF),
        "responses_equal_single_device": same,
        "note": "cpir_server_setup_multi + cpir_server_respond: one process, database split along the filter slots over the listed "
                "devices, query slices scattered over each device's own host link, C-word partial responses summed on the host; "
                "device_queries_*: cpir_server_respond_batch_device on the same handle (16 queries, one pass each): peer copies of the "
                "query slices, per-shard responds, partials pushed to the root and summed there by a kernel, all stream-ordered",
    }
    grp.close()
    return out


def sweep(cp, torch, run_step, qps_step):
    """time each respond kernel variant (one process, interleaved rounds) -- tuning aid, output on stderr"""
    variants = [(nt, bpc) for nt in (0, 1) for bpc in (1, 2, 3, 4, 0)]
    best = {}
    for rnd in range(3):
        for nt, bpc in variants:
            cp.tuning_set("respond.nontemporal", nt)
            cp.tuning_set("respond.blocks_per_cu", bpc)
            run_step()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                run_step()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / (3 * qps_step)
            best[(nt, bpc)] = min(best.get((nt, bpc), 1e30), us)
    for k, us in sorted(best.items(), key=lambda kv: kv[1]):
        log(f"sweep nt={k[0]} blocks/CU={k[1]}: {us:8.1f} us/query")
    nt, bpc = min(best, key=best.get)
    cp.tuning_set("respond.nontemporal", nt)
    cp.tuning_set("respond.blocks_per_cu", bpc)
    log(f"sweep: using nt={nt} blocks/CU={bpc}")


def cpu_ranges(cpus) -> str:
    """{0,1,2,3,8,9} -> '0-3,8-9'"""
    cpus = sorted(cpus)
    out, i = [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        out.append(str(cpus[i]) if i == j else f"{cpus[i]}-{cpus[j]}")
        i = j + 1
    return ",".join(out)


def host_cpu_description() -> dict:
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                if ln.startswith("model name"):
                    model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else round(int(q) / int(per), 2)
    except (OSError, ValueError):
        pass
    aff = os.sched_getaffinity(0)
    return {"cpu_model": model, "logical_cpus_visible": os.cpu_count(), "affinity_mask": cpu_ranges(aff), "affinity_cpus": len(aff), "cgroup_cpu_quota": quota}


def cpu_baseline_child(args) -> int:
    """--cpu-baseline-child DIR (internal): the oracle's restatement of the reference CPU path (matrix.rs:328-485: parallel over the C outputs,
    sequential fold per output) timed in a process of its own -- no torch, no GPU --, so that the OpenMP binding asked for in the
    environment (OMP_PROC_BIND / OMP_PLACES) binds THIS process's threads only and not the host threads of the library under test.
    Several samples (each a fixed share of the time budget, full-size queries back to back); the responses of the first queries are left
    in DIR for the parent to compare with the GPU's.  Prints one JSON object."""
    from oracle import oracle as orc  # baseline / checker only

    d = args.cpu_baseline_child
    with open(os.path.join(d, "meta.json")) as fh:
        meta = json.load(fh)
    orc.lib()
    cpus = orc.usable_cpus()
    orc.set_num_threads(cpus)
    N, b, nq = int(meta["N"]), int(meta["b"]), int(meta["queries"])
    if meta.get("generate"):
        # small configs: the child builds the database itself from the counter-based generator, through the oracle's own transpose + compress
        C = int(meta["C"])
        D = orc.synth_fill_u32(N * C, SEED_D, 0, (1 << b) - 1).reshape(N, C)
        dtc = orc.row_wise_compress(orc.transpose(D), b)
        del D
        qs = [orc.synth_fill_u32(N, SEED_Q + i) for i in range(nq)]
    else:
        dtc = np.load(os.path.join(d, "dtc.npy"), mmap_mode="r")
        qall = np.load(os.path.join(d, "q.npy"))
        qs = [qall[i] for i in range(nq)]
    dtc = orc.first_touch_copy(dtc)  # pages placed next to the threads that stream them (NUMA)
    wants = [orc.row_vector_x_compressed_transposed_matrix(q, dtc, N, b)[0] for q in qs]  # (also the warm-up)
    np.save(os.path.join(d, "responses.npy"), np.stack(wants))
    n_samples = max(3, int(meta.get("samples", 3)))
    per = float(meta["seconds"]) / n_samples
    samples, n_total = [], 0
    for _ in range(n_samples):
        n, t0 = 0, time.perf_counter()
        while True:
            orc.row_vector_x_compressed_transposed_matrix(qs[n % nq], dtc, N, b)
            n += 1
            el = time.perf_counter() - t0
            if el >= per or n >= 100000:
                break
        samples.append(n / el)
        n_total += n
    print(json.dumps({"samples_queries_per_sec": [round(x, 3) for x in samples], "queries_timed": n_total, "threads": cpus,
                      "omp_proc_bind": os.environ.get("OMP_PROC_BIND"), "omp_places": os.environ.get("OMP_PLACES"), **host_cpu_description()}), flush=True)
    return 0


def run_cpu_children(workdir, budget_s):
    """the CPU baseline three times, each in a child process: threads bound side by side (OMP_PROC_BIND=close OMP_PLACES=cores: comparable from
    box to box where the box lets the process have those cores -- but 16 threads side by side share two core complexes' links to memory),
    bound far apart (OMP_PROC_BIND=spread: one thread per core complex where there are enough, what a bandwidth-bound loop wants) and unbound
    (the scheduler places them: what a quota'd container on a shared host often does best with); returns {"bound": {...}, "spread": {...},
    "unbound": {...}} (an entry may hold "error")"""
    import subprocess

    out = {}
    for name, env_extra in (("bound", {"OMP_PROC_BIND": "close", "OMP_PLACES": "cores"}), ("spread", {"OMP_PROC_BIND": "spread", "OMP_PLACES": "cores"}),
                            ("unbound", {"OMP_PROC_BIND": "false"})):
        env = {k: v for k, v in os.environ.items() if k not in ("OMP_PLACES", "OMP_PROC_BIND", "OMP_NUM_THREADS", "GOMP_CPU_AFFINITY")}
        env.update(env_extra)
        env["OMP_WAIT_POLICY"] = "active"  # the child has its cores to itself while it runs (the parent waits, its device idle)
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", workdir], capture_output=True, text=True,
                               timeout=max(120.0, 20 * budget_s), env=env)
            lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            out[name] = json.loads(lines[-1]) if (p.returncode == 0 and lines) else {"error": f"child exited with {p.returncode}: {p.stderr[-300:]}"}
        except Exception as exc:  # noqa: BLE001
            out[name] = {"error": repr(exc)}
    return out


def summarize_cpu_runs(runs, full_bytes, sample_text):
    """the `cpu_baseline` object from the two child runs: value = the better of the two medians (the reference would be run the way that suits
    the box), every sample of both kept"""
    def stats(r):
        xs = sorted(r["samples_queries_per_sec"])
        return {"min": xs[0], "median": xs[len(xs) // 2], "max": xs[-1], "samples": r["samples_queries_per_sec"], "queries_timed": r["queries_timed"]}

    good = {k: stats(v) for k, v in runs.items() if "error" not in v}
    if not good:
        return {"error": {k: v.get("error") for k, v in runs.items()}}
    best = max(good, key=lambda k: good[k]["median"])
    any_run = runs[best]
    return {
        "value": good[best]["median"],
        "unit": "queries/s",
        "cores": any_run["threads"],
        "kind": "port",
        "sample": sample_text,
        "value_is": f"median of {len(good[best]['samples'])} samples, OpenMP threads {best} (the best median of the three placements)",
        "min": good[best]["min"], "median": good[best]["median"], "max": good[best]["max"],
        "threads_bound": good.get("bound") or runs.get("bound"),      # OMP_PROC_BIND=close OMP_PLACES=cores
        "threads_spread": good.get("spread") or runs.get("spread"),   # OMP_PROC_BIND=spread OMP_PLACES=cores
        "threads_unbound": good.get("unbound") or runs.get("unbound"),  # OMP_PROC_BIND=false
        "cpu_model": any_run["cpu_model"],
        "affinity_mask": any_run["affinity_mask"],
        "cgroup_cpu_quota": any_run["cgroup_cpu_quota"],
        "host": f"{any_run['cpu_model']}: {any_run['logical_cpus_visible']} logical CPUs visible, affinity mask {any_run['affinity_mask']} ({any_run['affinity_cpus']} CPUs), "
                f"cgroup CPU quota {any_run['cgroup_cpu_quota']}; {any_run['threads']} OpenMP threads",
        "GBps": round(full_bytes * good[best]["median"] / 1e9, 1),
        # the reference itself cannot be built here (Rust, no toolchain): what its authors publish for this bench on other hardware
        # (divan medians of `server_respond`, 2^20 keys x 1 kB, 3-wise filter)
        "reference_published_queries_per_sec": {"aws m8g.8xlarge (Graviton4, 32 vCPU)": 99.4, "aws r8g.8xlarge": 89.2, "aws m7i.8xlarge (x86_64)": 71.1},
    }


def cpu_workdir(need_bytes: int = 0):
    """a directory for the files the CPU baseline's child processes read: memory-backed where that has the room, else /tmp"""
    import shutil
    import tempfile

    base = "/tmp"
    try:
        if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) and shutil.disk_usage("/dev/shm").free > need_bytes + (256 << 20):
            base = "/dev/shm"
    except OSError:
        pass
    return tempfile.mkdtemp(prefix="cpir_cpu_", dir=base)


def cpu_baseline(server, q_pool, r_step, N, C, b, full_bytes, budget_s, torch, stream, config):
    """Oracle (C restatement of the reference's rayon CPU path, matrix.rs:328-485) on this box's host cores, same DB and queries, in CHILD
    processes (cpu_baseline_child): timed are back-to-back full-size CPU queries only, on as many OpenMP threads as the process may really use
    (affinity mask / cgroup CPU quota), three samples with the threads bound to cores and three with the scheduler placing them.  Then,
    untimed, the CPU responses are compared bit for bit with the GPU responses of the same queries."""
    import shutil

    nq = min(q_pool.shape[0], 32)
    work = cpu_workdir(full_bytes + 4 * N * nq)
    try:
        np.save(os.path.join(work, "dtc.npy"), server.export_compressed())  # the reference's own C x ceil(N / cf) words, exported from the device image
        np.save(os.path.join(work, "q.npy"), q_pool[:nq].cpu().numpy().view(np.uint32))
        with open(os.path.join(work, "meta.json"), "w") as fh:
            json.dump({"N": N, "b": b, "queries": nq, "seconds": budget_s / 3, "samples": 3}, fh)
        runs = run_cpu_children(work, budget_s)
        out = summarize_cpu_runs(runs, full_bytes, f"full-size queries of {config} back to back on the same packed DB ({full_bytes / 1e9:.3f} GB/query) for "
                                 f"{budget_s / 3:.0f} s per thread placement, OpenMP over the C outputs like the reference's rayon loop, rows first-touched by "
                                 "the threads that stream them")
        if "error" in out:
            return out
        wants = np.load(os.path.join(work, "responses.npy"))
    finally:
        shutil.rmtree(work, ignore_errors=True)
    mismatches = 0
    for i in range(nq):
        server.respond_device(q_pool[i], r_step[0], stream=stream)
        torch.cuda.synchronize()
        mismatches += int(not np.array_equal(r_step[0].cpu().numpy().view(np.uint32), wants[i]))
    out["gpu_results_bit_exact"] = mismatches == 0
    out["queries_compared"] = nq
    return out


def cpu_baseline_small(cp, device, torch, name, stream, budget_s=4.0):
    """the same CPU baseline at a small config (BASELINE.json configs[0], 2^16 keys: the reference's own CPU-runnable case): the child builds
    the database from the counter-based generator through the oracle's transpose + compress; this process builds the same database on the GPU
    and compares the responses of the same 32 queries"""
    import shutil

    n_keys, arity, value_bytes = CONFIGS[name]
    b = cp.find_encoded_db_matrix_element_bit_length(n_keys)
    _, _, N = cp.filter_shape(arity, n_keys)
    C = cp.encoded_num_cols(value_bytes, b)
    cf = 2 if b >= 11 else (3 if b >= 9 else 4)
    full_bytes = 4 * C * -(-N // cf) + 4 * N + 4 * C
    nq = 32
    work = cpu_workdir()
    try:
        with open(os.path.join(work, "meta.json"), "w") as fh:
            json.dump({"N": N, "C": C, "b": b, "queries": nq, "seconds": budget_s / 3, "samples": 3, "generate": True}, fh)
        runs = run_cpu_children(work, budget_s)
        out = summarize_cpu_runs(runs, full_bytes, f"full-size queries of {name} back to back ({full_bytes / 1e6:.1f} MB/query: last-level-cache sized on a "
                                 f"server CPU) for {budget_s / 3:.1f} s per thread placement, database built by the oracle's own transpose + row_wise_compress")
        if "error" in out:
            return out
        wants = np.load(os.path.join(work, "responses.npy"))
    finally:
        shutil.rmtree(work, ignore_errors=True)
    out["workload"] = workload_name(name, N, C, b, cf)
    out.pop("reference_published_queries_per_sec", None)
    D = torch.empty((N, C), dtype=torch.int32, device="cuda")
    device.synth_fill(D, N * C, SEED_D, mask=(1 << b) - 1, stream=stream)
    srv = cp.Server.from_device_matrix(D, N, C, b, device=device, stream=stream)
    q = torch.empty(N, dtype=torch.int32, device="cuda")
    r = torch.empty(C, dtype=torch.int32, device="cuda")
    mismatches = 0
    for i in range(nq):
        device.synth_fill(q, N, SEED_Q + i, stream=stream)
        srv.respond_device(q, r, stream=stream)
        torch.cuda.synchronize()
        mismatches += int(not np.array_equal(r.cpu().numpy().view(np.uint32), wants[i]))
    srv.close()
    del D, q, r
    torch.cuda.empty_cache()
    out["gpu_results_bit_exact"] = mismatches == 0
    out["queries_compared"] = nq
    return out


def setup_timing(cp, device, torch, sharded, N, C, b, mask, stream):
    """server_setup wall seconds from (seed_mu, encoded D on the host): XOF expansion of A (host) || D upload + pack,
    hint matmul, hint download -- Server::setup minus the KV encoder (reference server.rs:59-67)."""
    D_dev = torch.empty((N, C), dtype=torch.int32, device="cuda")
    device.synth_fill(D_dev, N * C, SEED_D, mask=mask, stream=stream)
    torch.cuda.synchronize()
    D_host = D_dev.cpu().numpy().view(np.uint32)
    del D_dev
    torch.cuda.empty_cache()
    t0 = time.perf_counter()
    srv, hint = cp.Server.setup_from_matrix(SEED_MU, D_host, b, device=device)
    wall = time.perf_counter() - t0
    phases = srv.setup_timings()
    # cheap integrity check: both servers hold the same packed DB => same response
    q = torch.empty(N, dtype=torch.int32, device="cuda")
    device.synth_fill(q, N, 0xABCDEF, stream=stream)
    r1 = torch.empty(C, dtype=torch.int32, device="cuda")
    r2 = torch.empty(C, dtype=torch.int32, device="cuda")
    srv.respond_device(q, r1, stream=stream)
    sharded.local.respond_device(q, r2, stream=stream)
    torch.cuda.synchronize()
    same = bool(torch.equal(r1, r2))
    macs = 1774 * N * C
    return {
        "server_setup_wall_sec": round(wall, 3),
        "server_setup_phases_sec": {k: round(v, 4) for k, v in phases.items()},
        "server_setup_note": "setup(seed_mu, encoded D on host): A expanded by TurboSHAKE128 on one host core (sequential sponge) "
                             "overlapped with D upload + transpose/pack and with the hint matmul of the rows already expanded",
        "hint_matmul_note": "the hint is computed in chunks of 128 rows of A as they arrive; `hint_matmul` is what is left after the last "
                            f"rows are there ({macs / 1e12:.2f} T multiply-adds in all: kernel time in profiles/*setup_kernel_stats.csv)",
        "setup_db_matches_bench_db": same,
        "hint_checksum": int(hint.sum(dtype=np.uint64) & 0xFFFFFFFFFFFFFFFF),
    }


VALU_DOT2_PEAK_TMACS = 37.3  # measured issue rate of v_dot2_u32_u16 on MI355X, one lane-op per u32 MAC (scripts/valu_rate.hip)
MFMA_I8_PEAK_TOPS = 5000.0   # dense i8 MFMA = 2 x the bf16 rate (MI355X_MICROARCH.md, Matrix cores): ~5 POP/s = 2.5e15 i8 MACs/s


def setup_kernel_roofline(cp, device, torch, N, C, b, cf, mask, layout, stream):
    """the two offline kernels against their roofs, each timed alone with HIP events on the launch stream, inputs resident
    in HBM (synthetic A and D of the config's shape): the hint matmul (1774 x N by N x C, u32 wrap-around) and transpose + compress."""
    import ctypes

    R = 1774
    D = torch.empty((N, C), dtype=torch.int32, device="cuda")
    device.synth_fill(D, N * C, SEED_D, mask=mask, stream=stream)
    A = torch.empty((R, N), dtype=torch.int32, device="cuda")
    device.synth_fill(A, R * N, 0xA, stream=stream)
    M = torch.empty((R, C), dtype=torch.int32, device="cuda")
    dtc = torch.empty(int(layout.total_words) + int(layout.rows_padded) + 64, dtype=torch.int32, device="cuda")

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    mm_ms = timed(lambda: device.mat_x_mat(A, D, M, R, N, C, rhs_max_bits=16, stream=stream))
    pk_ms = timed(lambda: device.transpose_compress(D, layout, dtc, stream=stream))
    # what Server::setup runs where it can (planar packing, b >= 9): ONE pass over D writes the packed image and the second operand plane
    # of the matmul, whose first plane are the image's own low-byte pieces -- no separate split of D in front of the product
    paired = None
    plane_bytes = cp.packed_rhs_plane_bytes(layout)  # 0 with one bit plane (b = 9): the matmul expands it from the image itself
    if cp.packed_rhs_offered(layout) and "mfma" in cp.mat_x_mat_kernel_name(16):
        plane = torch.empty(plane_bytes // 4, dtype=torch.int32, device="cuda") if plane_bytes else None
        M2 = torch.empty((R, C), dtype=torch.int32, device="cuda")
        pk2_ms = timed(lambda: device.transpose_compress_with_plane(D, layout, dtc, plane, stream=stream))
        mm2_ms = timed(lambda: device.mat_x_packed(A, dtc, layout, plane, M2, R, stream=stream))
        paired = {"pack_with_plane_ms": round(pk2_ms, 3), "matmul_ms": round(mm2_ms, 3), "same_hint_as_split_path": bool(torch.equal(M, M2)),
                  "plane_bytes": plane_bytes}
        split_ms, mm_ms = mm_ms, mm2_ms
        pack_alone_ms, pk_ms = pk_ms, pk2_ms
        del plane, M2
    macs = R * N * C
    b_setup = 4 * R * N + 4 * N * C + 4 * R * C
    b_pack_alg = 4 * N * C + 4 * C * -(-N // cf)
    b_pack_moved = 4 * N * C + 4 * int(layout.total_words)
    kernel = cp.mat_x_mat_kernel_name(16)
    mfma = "mfma" in kernel
    tmacs = macs / (mm_ms * 1e-3) / 1e12
    out = {
        "hint_matmul": {
            "kernel": kernel,
            "ms": round(mm_ms, 3),
            "u32_TMACs_per_s": round(tmacs, 2),
            "bound": "mfma" if mfma else "valu",
            # one u32 x (<= 16-bit) wrap-around MAC = 4 x 2 signed-byte products, of which the kernel ISSUES 7 (the limb pair with
            # i + j = 4 only reaches bits >= 32 and is never computed: matmul_mfma.hip); `achieved` / `frac` count the executed ops
            # (2 per byte MAC), the 8-based figure is kept as "algorithmic".  On the VALU: one v_dot2_u32_u16 lane-op per u32 MAC.
            "achieved": round(tmacs * 14, 1) if mfma else round(tmacs, 2),
            "peak": MFMA_I8_PEAK_TOPS if mfma else VALU_DOT2_PEAK_TMACS,
            "unit": "i8 TOP/s executed (7 i8 MACs issued per u32 MAC)" if mfma else "T lane-ops/s (v_dot2_u32_u16, one per u32 MAC)",
            "frac": round(tmacs * 14 / MFMA_I8_PEAK_TOPS, 4) if mfma else round(tmacs / VALU_DOT2_PEAK_TMACS, 4),
            "algorithmic_i8_TOPs": round(tmacs * 16, 1) if mfma else None,  # 8 byte products per u32 MAC, as if none vanished
            "algorithmic_bytes": b_setup,
            "hbm_GBps": round(b_setup / (mm_ms * 1e-3) / 1e9, 1),
            "hbm_frac": round(b_setup / (mm_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
        },
        "transpose_compress": {
            "kernel": cp.pack_kernel_name(layout),  # as the kernel trace shows it
            "ms": round(pk_ms, 3),
            "bound": "hbm",
            "algorithmic_bytes": b_pack_alg,
            # a rate: the bytes this pass really reads and writes (D once + the resident image) over its time; SURVEY.md 8(d)'s figure -- the
            # reference packing's bytes -- beside it
            "achieved": round(b_pack_moved / (pk_ms * 1e-3) / 1e9, 1),
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": round(b_pack_moved / (pk_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
            "frac_algorithmic_equiv": round(b_pack_alg / (pk_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
            "moved_bytes": b_pack_moved,
            "moved_GBps": round(b_pack_moved / (pk_ms * 1e-3) / 1e9, 1),
        },
        "note": "each kernel alone, one launch over the whole config, HIP events on the launch stream, synthetic A / D resident in HBM; "
                "inside Server::setup both hide behind the host XOF (server_setup_phases_sec)",
    }
    if paired:
        out["hint_matmul"]["kernel"] = "mat_x_mat_mfma_pipe_kernel<image + bit plane>" if not paired["plane_bytes"] else "mat_x_mat_mfma_pipe_kernel<image + byte plane>"
        out["hint_matmul"]["right_hand_side"] = (
            "the packed image's low-byte pieces + its one bit plane expanded to the high-byte operand in registers (cpir_op_mat_x_packed): "
            "D is read once and the pack pass writes nothing but the image" if not paired["plane_bytes"] else
            "the packed image's low-byte pieces + the high-byte plane written by the pack pass "
            "(cpir_op_transpose_compress_with_plane + cpir_op_mat_x_packed): D is read once")
        out["hint_matmul"]["same_hint_as_split_path"] = paired["same_hint_as_split_path"]
        out["hint_matmul"]["split_path_ms"] = round(split_ms, 3)  # cpir_op_mat_x_mat: byte-plane split of D (a pass of its own) + product
        out["transpose_compress"]["writes_matmul_plane_bytes"] = paired["plane_bytes"]
        out["transpose_compress"]["moved_bytes"] = b_pack_moved + paired["plane_bytes"]
        out["transpose_compress"]["moved_GBps"] = round((b_pack_moved + paired["plane_bytes"]) / (pk_ms * 1e-3) / 1e9, 1)
        out["transpose_compress"]["achieved"] = out["transpose_compress"]["moved_GBps"]
        out["transpose_compress"]["frac"] = round(out["transpose_compress"]["moved_GBps"] / HBM_PEAK_GBPS, 4)
        out["transpose_compress"]["without_plane_ms"] = round(pack_alone_ms, 3)
    del A, D, M, dtc
    torch.cuda.empty_cache()
    # the pack pass moves a 3.55 : 1 mix of reads and writes; what a plain copy-like kernel of that mix reaches on THIS device, measured now
    # (a child process, after the buffers above are gone), is its ceiling -- 8 TB/s is the data sheet's read+write peak, which no kernel sees
    mix = mix_ceiling(min(4 * N * C, 16 << 30))
    if mix:
        tc = out["transpose_compress"]
        tc["copy_ceiling_GBps"] = mix["mix_ceiling_GBps"]
        tc["frac_vs_copy_ceiling"] = round(tc["moved_GBps"] / mix["mix_ceiling_GBps"], 4)
        tc["copy_ceiling_note"] = mix["kernel"]
    return out


def setup_timing_sharded(cp, device, torch, dist, N, C, b, mask, lo, hi, rank, stream, unit):
    """server_setup on the N-sharded database with ONE expansion of A per node: rank 0 squeezes the sponge block by block (it is
    sequential, so this is the floor of setup whatever the number of GPUs), uploads each block once and BROADCASTS it over the process
    group (RCCL over xGMI; every rank cuts its column slab out of the landed block -- host-staged test backends send slabs point to point
    instead: chalametpir_amd/distributed.py::scatter_public_matrix); every rank multiplies its slab with its shard of D on the matrix cores and packs its
    shard; the partial hints are sum-reduced to rank 0.  Wall time = max over ranks, barrier to barrier."""
    from chalametpir_amd.distributed import reduce_u32_, scatter_public_matrix

    D_dev = torch.empty(((hi - lo), C), dtype=torch.int32, device="cuda")
    if hi > lo:
        device.synth_fill(D_dev, (hi - lo) * C, SEED_D, index0=lo * C, mask=mask, stream=stream)
    M = torch.zeros((1774, C), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    if hi > lo:  # the shard is packed while the sponge is still being squeezed
        srv = cp.Server.from_device_matrix(D_dev, hi - lo, C, b, device=device, slot_offset=lo, total_slots=N, stream=stream)
    slab, lo2, hi2 = scatter_public_matrix(SEED_MU, N, unit, device=torch.device("cuda", torch.cuda.current_device()))
    assert (lo2, hi2) == (lo, hi)
    t_scatter = time.perf_counter() - t0
    if hi > lo:
        device.mat_x_mat(slab, D_dev, M, 1774, hi - lo, C, rhs_max_bits=16, stream=stream)
    torch.cuda.synchronize()
    t_local = time.perf_counter() - t0
    reduce_u32_(M, dst=0)
    torch.cuda.synchronize()
    dist.barrier()
    wall = time.perf_counter() - t0
    t = torch.tensor([wall, t_local, t_scatter], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    out = {
        "server_setup_wall_sec": round(float(t[0].item()), 3),
        "server_setup_note": "sharded setup from (seed_mu, encoded D shards in HBM), ONE XOF expansion of A per node: rank 0 squeezes, uploads and "
                             "broadcasts block by block (RCCL; host-staged test backends: slabs point to point), every rank keeps its column slab: "
                             "shard pack, partial hint matmul (matrix cores) on its slab; partial hints sum-reduced to rank 0",
        "server_setup_max_rank_local_sec": round(float(t[1].item()), 3),
        "server_setup_expand_and_scatter_sec": round(float(t[2].item()), 3),
        "xof_expansions_per_node": 1,
    }
    if rank == 0:
        hint = M.cpu().numpy().view(np.uint32)
        out["hint_checksum"] = int(hint.sum(dtype=np.uint64) & 0xFFFFFFFFFFFFFFFF)
    return out


def synthetic_kv_database(n_keys, value_bytes):
    """n distinct 32-byte keys and n values (seeded): the flat arrays of cpir_kv_db"""
    rng = np.random.default_rng(0xC0FFEE)
    keys = rng.integers(0, 256, size=n_keys * 32, dtype=np.uint8)
    keys.reshape(n_keys, 32)[:, :8] = np.arange(n_keys, dtype=np.uint64).view(np.uint8).reshape(n_keys, 8)  # distinct keys
    values = rng.integers(0, 256, size=n_keys * value_bytes, dtype=np.uint8)
    key_off = np.arange(n_keys + 1, dtype=np.uint64) * 32
    val_off = np.arange(n_keys + 1, dtype=np.uint64) * value_bytes
    return keys, key_off, values, val_off


def setup_kv_and_real_db(cp, device, torch, args, n_keys, arity, value_bytes, q_pool, N, C, b, cf, stream):
    """(1) The reference's `server_setup` bench (integrations/benches/offline_phase.rs:59-72): Server::setup::<ARITY>(seed, db) with the KV
    database as input -- filter construction + row encoding + A expansion + hint + packed DB all timed.
    (2) `real_db`: the headline's timed loop (one query per pass, every pass its own stream of the database) on THAT server -- a database
    as the binary fuse filter really encodes it: N - n of its N rows belong to no key and are all zero (matrix.rs:702-746).
    (3) a few keys looked up end to end through the oracle's client restatement (checker only): query -> respond on wire bytes -> decode."""
    keys, key_off, values, val_off = synthetic_kv_database(n_keys, value_bytes)
    t0 = time.perf_counter()
    srv, hint_bytes, filter_bytes = cp.Server.setup_flat(SEED_MU, keys, key_off, values, val_off, arity, device=device)
    wall = time.perf_counter() - t0
    phases = srv.setup_timings()
    out = {
        "server_setup_kv_wall_sec": round(wall, 3),
        "server_setup_kv_phases_sec": {k: round(v, 4) for k, v in phases.items()},
        "server_setup_kv_note": f"Server::setup::<{arity}>(seed, {n_keys} x (32 B, {value_bytes} B)): BFF construction + row encoding on host "
                                "threads || TurboSHAKE128 expansion of A on one host thread, then D upload, pack, hint matmul",
        "hint_bytes": len(hint_bytes),
        "filter_param_bytes": len(filter_bytes),
    }
    try:
        assert srv.decompressed_num_cols == N and srv.mat_elem_bit_len == b
        qps_step, pool = args.queries_per_step, q_pool.shape[0]
        r = torch.zeros((qps_step, C), dtype=torch.int32, device="cuda")
        cp.tuning_set("respond.batch_fusion", 0)

        def step(k):
            base = (k * qps_step) % pool
            srv.respond_batch_device(q_pool[base:base + qps_step], qps_step, r, stream=stream)

        for k in range(max(2, args.warmup // 2)):
            step(k)
        torch.cuda.synchronize()
        n_steps = max(4, args.steps // 2)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for k in range(n_steps):
            step(k)
        e1.record(stream)
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (n_steps * qps_step)
        full_bytes = 4 * C * -(-N // cf) + 4 * N + 4 * C
        resident = int(srv.physical_layout.total_words) * 4
        served, of = srv.slots_served()
        real = {
            "queries_per_sec": round(1e6 / us, 1),
            "us_per_query": round(us, 2),
            "rows_owned_by_no_key": N - n_keys,
            "rows_owned_by_no_key_frac": round((N - n_keys) / N, 4),
            "slots_served": served,
            "slots_of": of,
            "resident_bytes": resident,
            # the physical figure, a rate: the compact image + the slot map and the query words of the kept slots (gathered inside the kernel)
            # + r, over the time, against 8 TB/s
            "frac": None if resident <= (256 << 20) else round((resident + 8 * served + 4 * C) / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4),
            "frac_moved": None if resident <= (256 << 20) else round((resident + 8 * served + 4 * C) / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4),
            "moved_GBps": round((resident + 8 * served + 4 * C) / (us * 1e-6) / 1e9, 1),
            "mall_resident": bool(resident <= (256 << 20)),  # (then the bytes are served on die and no fraction of the HBM roof is stated)
            # NOT a bandwidth: the algorithmic bytes of ALL N slots in the reference packing over the time of a kernel that
            # streams only the kept ones in a tighter packing -- it may exceed 1.0; how fast a kernel streaming the reference's bytes would
            # have to be to answer as quickly
            "frac_algorithmic_equiv": None if resident <= (256 << 20) else round(full_bytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4),
            "note": "the headline's loop (uniform random queries -- what an LWE query is to the server --, one query per pass, "
                    f"{qps_step} passes a launch) on the server that Server::setup built from the key-value database; `frac` (= `frac_moved`) is "
                    "bytes through HBM over time",
        }
        if pool >= 64:  # the same database with fused batches, 48 queries a launch (planar: two wide passes of 24)
            cp.tuning_set("respond.batch_fusion", 1)
            rb = torch.zeros((48, C), dtype=torch.int32, device="cuda")
            for k in range(12):
                srv.respond_batch_device(q_pool[16 * (k % 2):16 * (k % 2) + 48], 48, rb, stream=stream)
            torch.cuda.synchronize()
            e0.record(stream)
            for k in range(10):
                srv.respond_batch_device(q_pool[16 * (k % 2):16 * (k % 2) + 48], 48, rb, stream=stream)
            e1.record(stream)
            torch.cuda.synchronize()
            real["fused_us_per_query"] = round(e0.elapsed_time(e1) * 1e3 / (10 * 48), 2)
            real["fused_queries_per_pass"] = cp.respond_batch_pass_width(srv.physical_layout, 48)
            cp.tuning_set("respond.batch_fusion", 0)
            del rb
        if not args.no_host_path:
            cp.tuning_set("respond.batch_fusion", 1)
            real["respond_host_path"] = host_path_timing(srv, q_pool, N, torch, full=False)
            cp.tuning_set("respond.batch_fusion", 0)
        if args.real_db_keys > 0:
            real["end_to_end"] = real_db_lookup(cp, srv, hint_bytes, filter_bytes, keys, values, n_keys, value_bytes, args.real_db_keys)
        out["real_db"] = real
    finally:
        srv.close()
    return out


def real_db_lookup(cp, srv, hint_bytes, filter_bytes, keys, values, n_keys, value_bytes, n_lookups):
    """keyword PIR end to end on the real database (reference integrations/src/test_pir.rs:12-142): the oracle's restatement of the client
    (checker only; client.rs:95-194, 209-275) builds LWE queries for a few keys, the GPU server answers them on wire bytes, the client decodes"""
    from oracle import oracle as orc  # checker only

    t0 = time.perf_counter()
    filt = orc.Filter.from_bytes(filter_bytes)
    rows, cols = np.frombuffer(hint_bytes[:8], dtype="<u4")
    hint = np.frombuffer(hint_bytes[8:], dtype="<u4").reshape(int(rows), int(cols))
    Nf = filt.num_fingerprints
    A = orc.generate_from_seed(1774, Nf, SEED_MU)
    rng = np.random.default_rng(0xE2E)
    idx = [0, n_keys - 1] + [int(x) for x in rng.integers(0, n_keys, size=max(0, n_lookups - 2))]
    idx = idx[:n_lookups]
    ind = orc.query_indicator(filt.mat_elem_bit_len)
    ok, tried = 0, 0
    S = np.stack([orc.ternary_vector(1774, rng) for _ in idx])
    B, Cc = orc.mul(S, A), orc.mul(S, hint)
    del A
    for j, i in enumerate(idx):
        key = keys[32 * i:32 * (i + 1)].tobytes()
        q = B[j] + orc.ternary_vector_np(Nf, rng)
        slots = orc.filter_slots(filt, key)
        if any(int(q[h]) + ind >= (1 << 32) for h in slots):
            continue  # ArithmeticOverflowAddingQueryIndicator: the reference's caller retries with a fresh secret (test_pir.rs:66-70)
        for h in slots:
            q[h] += np.uint32(ind)
        tried += 1
        resp = srv.respond(np.array([1, Nf], dtype="<u4").tobytes() + q.tobytes())
        got = orc.client_process_response(filt, key, Cc[j], np.frombuffer(resp[8:], dtype="<u4"))
        ok += int(got == values[value_bytes * i:value_bytes * (i + 1)].tobytes())
    return {"keys_looked_up": tried, "values_recovered": ok, "all_recovered": bool(tried > 0 and ok == tried),
            "seconds": round(time.perf_counter() - t0, 2),
            "note": "client = the oracle's restatement (checker only); server = cpir_server_respond_bytes on the database built by cpir_server_setup_kv"}


if __name__ == "__main__":
    sys.exit(main())
