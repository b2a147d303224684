#!/bin/bash
# Runs ON THE GPU BOX, diagnosis build (make diag): the fused pass of 24 with 12 and with 11 matrix instructions per k-block (CPIR_WIDE_ABLATE=128:
# the sixth row set without its high-plane MFMAs, responses WRONG -- the upper bound of what high-plane fragments with 3 rows per query
# could save): time per launch, SQ_INSTS_MFMA / matrix-core busy cycles (separate rocprofv3 --pmc pass, kernel trace only), shader clock and
# socket power under each loop.   Writes gpurun_out/r6/eleven_of_twelve.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r6; mkdir -p $O
OUT=$O/eleven_of_twelve.txt
: > $OUT
cd $ROOT
echo "# scripts/wide_ablate.py, masks 0 / 128 / 4 (us per launch; batch 48 = two passes of 24)" >> $OUT
CPIR_ABLATE_MASKS=0,128,0,128,4,0,128 timeout -k 10 400 python3 scripts/wide_ablate.py >> $OUT 2>&1 || exit 1
cat > /tmp/wide_loop_diag.py <<'PY'
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
sys.path.insert(0, ROOT)
import torch
from chalametpir_amd import _native
_native.use_diag_build()
import chalametpir_amd as cp
batch, launches = int(sys.argv[1]), int(sys.argv[2])
N, C, b = 1179648, 940, 9
dev = cp.Device(0); stream = torch.cuda.current_stream()
D = torch.empty((N, C), dtype=torch.int32, device="cuda")
dev.synth_fill(D, N * C, 0xD, mask=(1 << b) - 1, stream=stream)
srv = cp.Server.from_device_matrix(D, N, C, b, device=dev, stream=stream)
torch.cuda.synchronize(); del D
q = torch.empty((batch, N), dtype=torch.int32, device="cuda")
for i in range(batch): dev.synth_fill(q, N, 0x1000 + i, offset_words=i * N, stream=stream)
r = torch.empty((batch, C), dtype=torch.int32, device="cuda")
for _ in range(launches): srv.respond_batch_device(q, batch, r, stream=stream)
torch.cuda.synchronize()
PY
for mask in 0 128; do
  P=$O/pmc_11of12_$mask; rm -rf $P
  (cd /tmp && TMPDIR=/tmp CPIR_WIDE_ABLATE=$mask timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $P -o p -- python3 /tmp/wide_loop_diag.py 48 8 > $P.log 2>&1) || { echo "pmc pass failed (mask $mask)" >> $OUT; tail -5 $P.log >> $OUT; }
  python3 - $P $mask >> $OUT <<'PY'
import csv, glob, collections, sys
root, mask = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list); dur = []
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "respond_planar_wide_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "respond_planar_wide_kernel" in r["Kernel_Name"]: dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(f"# counters, CPIR_WIDE_ABLATE={mask}, launches of 48 (mean per dispatch): " + "  ".join(f"{k} {sum(v)/len(v):.4g} (n={len(v)})" for k, v in sorted(acc.items()))
      + (f"  kernel_us {sum(dur)/len(dur):.1f}" if dur else ""))
PY
  find $P -name "*.db" -delete 2>/dev/null
done
for mask in 0 128; do
  # (36 000 launches of 48 queries: ~18 s of the loop; the samples are taken from second 14 on)
  CPIR_WIDE_ABLATE=$mask timeout -k 10 90 bash scripts/probes/clock_under_load.sh "fused loop, CPIR_WIDE_ABLATE=$mask" python3 /tmp/wide_loop_diag.py 48 36000 >> $OUT 2>&1
done
cat $OUT
