#!/bin/bash
# Runs ON THE GPU BOX: SQ counter passes (separate rocprofv3 runs, kernel-trace only) over the planar pack kernel at cfg2.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/pmc_pack; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { n=$1; shift; timeout -k 10 250 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -o $n -- python3 $ROOT/scripts/setup_kernels_timing.py cfg2 2 > $O/$n.txt 2>&1; }
run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES
run p2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM
run p3 GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run p4 TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum
find $O -name "*.db" -delete
cd $ROOT && python3 - <<'PY'
import csv,glob,collections,os
root=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out","pmc_pack")
for p in ('p1','p2','p3','p4'):
    acc=collections.defaultdict(list)
    for f in glob.glob(f'{root}/{p}/**/*counter_collection.csv',recursive=True):
        for r in csv.DictReader(open(f)):
            if 'planar_pack' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items(): print(p,k,len(v),f"{sum(v)/len(v):.4g}")
PY
