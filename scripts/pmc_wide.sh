#!/bin/bash
# Runs ON THE GPU BOX: SQ counter passes (separate rocprofv3 runs, kernel-trace only) over the wide respond kernel: launches of 48 queries
# (two passes of 24, six row sets) and of 32 (two passes of 16, four row sets) at the 2^20-key x 1 kB shape.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/pmc_wide; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { n=$1; b=$2; shift; shift; timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -o $n -- python3 $ROOT/scripts/wide_loop.py $b 6 > $O/$n.txt 2>&1; }
for b in 48 32; do
  run p1_$b $b SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
  run p2_$b $b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE
  run p3_$b $b GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM_RD
done
find $O -name "*.db" -delete
cd $ROOT && python3 - <<'PY'
import csv,glob,collections,os
root=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out","pmc_wide")
for b in ('48','32'):
    for p in ('p1','p2','p3'):
        acc=collections.defaultdict(list)
        for f in glob.glob(f'{root}/{p}_{b}/**/*counter_collection.csv',recursive=True):
            for r in csv.DictReader(open(f)):
                if 'respond_planar_wide_kernel' in r['Kernel_Name']:
                    acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in acc.items(): print(f"batch {b}",p,k,len(v),f"{sum(v)/len(v):.4g}")
        dur=[]
        for f in glob.glob(f'{root}/{p}_{b}/**/*kernel_trace.csv',recursive=True):
            for r in csv.DictReader(open(f)):
                if 'respond_planar_wide_kernel' in r['Kernel_Name']:
                    dur.append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
        if dur: print(f"batch {b}",p,"kernel_us(mean of %d)"%len(dur),f"{sum(dur)/len(dur):.1f}")
PY
