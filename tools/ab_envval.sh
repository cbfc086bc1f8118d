#!/bin/bash
# GPU box: alternate bench runs over values of one environment variable:  bash tools/ab_envval.sh VAR "v1 v2 ..." [rounds]   ("-" = unset)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; OUT=gpurun_out/ab_$1.txt; rm -f $OUT
for i in $(seq 1 ${3:-3}); do
  for v in $2; do
    if [ "$v" = "-" ]; then unset $1; else export $1=$v; fi
    python bench.py --steps 30 --warmup 5 --single-dtype --no-roofline --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1=$v', d['value'], d['outputs_finite'])" | tee -a $OUT
  done
done
