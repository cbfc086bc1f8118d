#!/bin/bash
# GPU box: tracked tile table vs a candidate ($1), alternating; prints the one-scene and the batched (4 scenes) rate.
#   bash tools/ab_table.sh dualdiff_amd/tuned/_probe.json [rounds] [tag]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; OUT=gpurun_out/ab_table_${3:-x}.txt; rm -f $OUT
for i in $(seq 1 ${2:-3}); do
  for t in tracked candidate; do
    if [ $t = candidate ]; then export DD_TUNE_TABLE=$GRAFT_REPO_ROOT/$1; else unset DD_TUNE_TABLE; fi
    python bench.py --steps 50 --warmup 5 --single-dtype --no-roofline --no-cpu-baseline 2>>gpurun_out/ab_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t', round(d['value'],2), round(d.get('batched',{}).get('value',0),2), d['outputs_finite'])" | tee -a $OUT
  done
done
