#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + FETCH_SIZE / WRITE_SIZE counter passes) into a short summary.
Usage: summarize_rocprof.py <prof dir> <tag>   -> prints the summary and writes <prof dir>/summary.json"""
import csv
import glob
import json
import os
import sys


def find(root, pattern):
    return sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))


def kernel_stats(root):
    rows = []
    for f in find(root, "*kernel_stats.csv"):
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    return rows


def counters(root):
    """-> {kernel_name: {counter: [values per dispatch]}}"""
    out = {}
    for f in find(root, "*counter_collection.csv"):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = r.get("Kernel_Name") or r.get("Kernel Name") or ""
                c = r.get("Counter_Name") or r.get("Counter Name") or ""
                v = float(r.get("Counter_Value") or r.get("Counter Value") or 0)
                out.setdefault(k, {}).setdefault(c, []).append(v)
    return out


def is_respond(name):
    """the dominant online kernel: respond_kernel<...> (VALU path) or respond_planar_wide_kernel<...> (matrix-core path)"""
    return "respond_kernel" in name or "respond_planar_wide_kernel" in name


def traffic_record(root, summary, tag, git_head=""):
    """profiles/respond_traffic.json: HBM bytes per pass from the two counter passes, with the guide's gfx950 correction
    (FETCH_SIZE counts 64 B per 128 B request of a 16 B/lane coalesced stream -> x2; WRITE_SIZE as is), next to the byte
    counts the bench run under the profiler reports for the same launch"""
    r = summary.get("respond", {})
    if "FETCH_SIZE_mean_raw" not in r:
        return None
    try:
        with open(os.path.join(root, "fetch_bench.json")) as fh:
            bench = json.loads(fh.read().strip().splitlines()[-1])
    except (OSError, ValueError, IndexError):
        return None
    roof = bench["roofline"]
    passes = roof["passes_per_launch"]
    traffic = (2 * r["FETCH_SIZE_mean_raw"] + r.get("WRITE_SIZE_mean_raw", 0.0)) * 1024 / passes
    algo = roof["bytes_per_launch"] / passes
    layout = roof["moved_bytes_per_launch"] / passes
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import kernel_source_sha256  # fingerprint of the respond kernels' sources as they were when this was measured

    return {
        "config": bench["config"]["workload"].split(":")[0],
        "git_head": git_head,
        "kernel_source_sha256": kernel_source_sha256(),
        "kernel": r.get("name", ""),
        "workload": bench["config"]["workload"],
        "packing": roof["packing"].split(" ")[0],
        "pass_order": roof.get("pass_order", ""),
        "dispatches": r.get("FETCH_SIZE_dispatches"),
        "passes_per_launch": passes,
        "FETCH_SIZE_mean_KiB_per_launch": r["FETCH_SIZE_mean_raw"],
        "WRITE_SIZE_mean_KiB_per_launch": r.get("WRITE_SIZE_mean_raw"),
        "traffic_bytes_per_pass": traffic,
        "algorithmic_bytes_per_pass": algo,
        "layout_bytes_per_pass": layout,
        "traffic_over_algorithmic": traffic / algo,
        "traffic_over_layout_bytes": traffic / layout,
        "correction": "gfx950: FETCH_SIZE counts 64 B per 128 B request for 16 B/lane coalesced streaming reads -> x2 "
                      "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE taken as is",
        "source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), scripts/profile_gpu.sh {tag}; "
                  f"one launch = {passes} passes (queries)",
    }


def main():
    root, tag = sys.argv[1], sys.argv[2]
    git_head = sys.argv[3] if len(sys.argv) > 3 else ""
    summary = {"tag": tag, "kernels": [], "respond": {}}
    print(f"== rocprofv3 summary [{tag}] ==")
    stats = kernel_stats(os.path.join(root, "trace"))
    stats.sort(key=lambda r: -float(r.get("TotalDurationNs", r.get("Total Duration (ns)", 0)) or 0))
    print("-- kernel stats (rocprofv3 --kernel-trace --stats) --")
    for r in stats[:12]:
        name = r.get("Name", "")
        calls = int(float(r.get("Calls", 0)))
        avg = float(r.get("AverageNs", r.get("Average (ns)", 0)) or 0)
        tot = float(r.get("TotalDurationNs", r.get("Total Duration (ns)", 0)) or 0)
        pct = r.get("Percentage", "")
        print(f"{calls:7d} calls  avg {avg / 1e3:10.2f} us  total {tot / 1e6:10.3f} ms  {pct:>7}%  {name[:110]}")
        summary["kernels"].append({"name": name, "calls": calls, "avg_us": avg / 1e3, "total_ms": tot / 1e6})
        if is_respond(name) and "avg_us" not in summary["respond"]:
            summary["respond"].update({"name": name, "calls": calls, "avg_us": avg / 1e3})
    # the bench line printed inside the traced run: its HIP-event launch time must agree with the trace's average for that kernel
    try:
        with open(os.path.join(root, "trace_bench.json")) as fh:
            line = json.loads(fh.read().strip().splitlines()[-1])
        roof = line["roofline"]
        if "avg_us" in summary["respond"]:
            avg = summary["respond"]["avg_us"]
            algo, moved = roof["bytes_per_launch"], roof["moved_bytes_per_launch"]
            print(f"-- bench line inside this trace: value {line['value']} q/s, roofline.launch_us {roof['launch_us']} (HIP events) vs kernel-trace "
                  f"average {avg:.2f} us over {summary['respond']['calls']} dispatches ({(roof['launch_us'] / avg - 1) * 100:+.2f} %)")
            print(f"-- from the trace's average: {moved / avg / 1e3:.1f} GB/s moved = roofline.frac {moved / avg / 1e3 / 8000:.4f} of 8 TB/s; "
                  f"algorithmic-bytes equivalent (reference packing, SURVEY 8d) {algo / avg / 1e3:.1f} GB/s = {algo / avg / 1e3 / 8000:.4f}")
            # `frac` is the rate of bytes that move (bench.py respond_roofline); SURVEY 8(d)'s algorithmic-bytes figure beside it
            summary["respond"].update({"bench_launch_us": roof["launch_us"], "frac_from_trace_avg": moved / avg / 1e3 / 8000,
                                       "frac_moved_from_trace_avg": moved / avg / 1e3 / 8000,
                                       "frac_algorithmic_equiv_from_trace_avg": algo / avg / 1e3 / 8000})
    except (OSError, ValueError, IndexError, KeyError):
        pass
    full = kernel_stats(os.path.join(root, "full_trace"))
    full.sort(key=lambda r: -float(r.get("TotalDurationNs", 0) or 0))
    if full:
        print("-- kernel stats of the default run's other sections too (fused batches, lone launches, real database, host path: one kernel name, several launch shapes) --")
        summary["all_sections_kernels"] = []
        for r in full[:8]:
            name, calls = r.get("Name", ""), int(float(r.get("Calls", 0)))
            avg, tot = float(r.get("AverageNs", 0) or 0), float(r.get("TotalDurationNs", 0) or 0)
            print(f"{calls:7d} calls  avg {avg / 1e3:10.2f} us  total {tot / 1e6:10.3f} ms  {name[:110]}")
            summary["all_sections_kernels"].append({"name": name, "calls": calls, "avg_us": avg / 1e3, "total_ms": tot / 1e6})
    for label, sub, ctr in (("fetch", "pmc_fetch", "FETCH_SIZE"), ("write", "pmc_write", "WRITE_SIZE")):
        c = counters(os.path.join(root, sub))
        want = summary["respond"].get("name")  # the instantiation that dominates the kernel trace (the headline launches)
        for k, d in c.items():
            if is_respond(k) and ctr in d and (want is None or k == want):
                vals = d[ctr]
                mean = sum(vals) / len(vals)
                # FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of 1024 B by rocprofv3's derived metric
                summary["respond"][ctr + "_mean_raw"] = mean
                summary["respond"][ctr + "_dispatches"] = len(vals)
                print(f"-- {ctr}: respond_kernel mean over {len(vals)} dispatches = {mean:.1f} (raw units; x1024 = bytes => {mean * 1024 / 1e9:.4f} GB)")
    setup = kernel_stats(os.path.join(root, "setup_trace"))
    setup.sort(key=lambda r: -float(r.get("TotalDurationNs", 0) or 0))
    if setup:
        print("-- kernel stats of a run that includes Server::setup (hint matmul, transpose+pack) --")
        summary["setup_kernels"] = []
        for r in setup[:8]:
            name, calls = r.get("Name", ""), int(float(r.get("Calls", 0)))
            avg, tot = float(r.get("AverageNs", 0) or 0), float(r.get("TotalDurationNs", 0) or 0)
            print(f"{calls:7d} calls  avg {avg / 1e3:10.2f} us  total {tot / 1e6:10.3f} ms  {name[:110]}")
            summary["setup_kernels"].append({"name": name, "calls": calls, "avg_us": avg / 1e3, "total_ms": tot / 1e6})
    tr = traffic_record(root, summary, tag, git_head)
    if tr:
        print(f"-- HBM traffic per pass: {tr['traffic_bytes_per_pass'] / 1e9:.4f} GB = {tr['traffic_over_algorithmic']:.3f} x algorithmic, "
              f"{tr['traffic_over_layout_bytes']:.3f} x the bytes of the resident layout")
        with open(os.path.join(root, "respond_traffic.json"), "w") as fh:
            json.dump(tr, fh, indent=1)
    with open(os.path.join(root, "summary.json"), "w") as fh:
        json.dump(summary, fh, indent=1)


if __name__ == "__main__":
    main()
