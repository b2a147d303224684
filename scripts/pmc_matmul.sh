#!/bin/bash
# Runs ON THE GPU BOX: SQ counter passes (separate rocprofv3 runs, kernel-trace only) over the hint matmul kernel at cfg2.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/pmc_mm; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { n=$1; shift; timeout -k 10 250 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -o $n -- python3 $ROOT/scripts/setup_kernels_timing.py cfg2 2 > $O/$n.txt 2>&1; }
run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
run p2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE
run p3 GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM_RD
find $O -name "*.db" -delete
cd $ROOT && python3 - <<'PY'
import csv,glob,collections,os
root=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out","pmc_mm")
for p in ('p1','p2','p3'):
    acc=collections.defaultdict(list)
    for f in glob.glob(f'{root}/{p}/**/*counter_collection.csv',recursive=True):
        for r in csv.DictReader(open(f)):
            if 'mat_x_mat_mfma' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items(): print(p,k,len(v),f"{sum(v)/len(v):.4g}")
PY
