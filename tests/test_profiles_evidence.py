"""What the committed evidence of THIS round (profiles/r6_*) may and may not say: a field named `frac` is a rate of bytes that move through
HBM over 8 TB/s -- it cannot exceed 1 -- and lines whose bytes are served on die carry no fraction at all (bench.py respond_roofline);
the traffic records the bench quotes when rocprofv3 is absent name kernels that still exist."""
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def walk(o, path, out):
    if isinstance(o, dict):
        for k, v in o.items():
            if k in ("frac", "frac_moved") and v is not None:
                out.append((path + "/" + k, v))
            walk(v, path + "/" + k, out)
    elif isinstance(o, list):
        for i, v in enumerate(o):
            walk(v, f"{path}[{i}]", out)


def test_no_fraction_of_the_hbm_roof_above_one_in_this_rounds_lines():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r6_*.json")))
    assert files
    seen = 0
    for f in files:
        with open(f) as fh:
            text = fh.read()
        try:
            docs = [json.loads(text)]  # one document (the rocprof summaries)
        except ValueError:
            docs = [json.loads(line) for line in text.splitlines() if line.startswith("{")]  # bench output: one JSON object per line
        for doc in docs:
            found = []
            walk(doc, "", found)
            seen += len(found)
            bad = [(p, v) for p, v in found if not (0 <= v <= 1.0)]
            assert not bad, (os.path.basename(f), bad)
    assert seen > 10


def test_committed_traffic_records_name_kernels_of_this_source():
    with open(os.path.join(ROOT, "profiles", "respond_traffic.json")) as fh:
        doc = json.load(fh)
    with open(os.path.join(ROOT, "chalametpir_amd", "csrc", "respond_planar.hip")) as fh:
        src = fh.read()
    assert {r["config"] for r in doc["records"]} >= {"cfg2", "cfg3", "cfg4", "cfg5"}
    for r in doc["records"]:
        name = re.search(r"(respond_\w+)<", r["kernel"]).group(1)
        assert re.search(r"\b" + name + r"\(const PlanarArgs a\)", src), (r["config"], name)  # the kernel is still defined
        assert 0.99 < r["traffic_over_layout_bytes"] < 1.02 and 0.8 < r["traffic_over_algorithmic"] < 0.9, r["config"]
