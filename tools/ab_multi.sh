#!/bin/bash
# GPU box: alternate bench runs over several "VAR=val,VAR2=val2" settings ("-" = nothing set):
#   bash tools/ab_multi.sh "<setting> <setting> ..." [rounds] [tag]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; OUT=gpurun_out/ab_${3:-multi}.txt; rm -f $OUT
for i in $(seq 1 ${2:-3}); do
  for cfg in $1; do
    if [ "$cfg" = "-" ]; then E=""; else E=$(echo $cfg | tr ',' ' '); fi
    env $E python bench.py --steps 50 --warmup 5 --single-dtype --no-roofline --no-cpu-baseline 2>>gpurun_out/ab_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', round(d['value'],2), d['outputs_finite'])" | tee -a $OUT
  done
done
