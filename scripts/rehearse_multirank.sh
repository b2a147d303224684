#!/bin/bash
# Rehearsal of the driver's scaling command on a ONE-GPU box: `python3 bench.py --gpus R --steps 20 --warmup 5 [--config ...]` with R ranks
# sharing GPU 0 and gloo as the collective (RCCL refuses two ranks on one device).  The pool allows at most 6 processes on a card, and the launcher (torch.distributed.run) counts as one of them, so R <= 5.
# (Nothing else may open the card meanwhile: a rocm-smi monitor counted as the 7th process.)  Writes the printed lines and the wall time to gpurun_out/r6/rehearsal_<config>_<R>ranks.{json,log}.
set -u
R=${1:-5}; shift
mkdir -p gpurun_out/r6
for cfg in "$@"; do
  out=gpurun_out/r6/rehearsal_${cfg}_${R}ranks
  t0=$(date +%s)
  CPIR_BENCH_BACKEND=gloo CPIR_BENCH_SHARE_DEVICE=1 OMP_NUM_THREADS=2 timeout -k 10 1000 python3 bench.py --gpus $R --steps 20 --warmup 5 --config $cfg > $out.json 2> $out.log
  rc=$?
  t1=$(date +%s)
  echo "{\"rehearsal\": \"$cfg\", \"ranks\": $R, \"exit_code\": $rc, \"wall_sec\": $((t1 - t0))}" >> $out.json
  tail -2 $out.json | cut -c1-600
  [ $rc -eq 0 ] || exit $rc
done
