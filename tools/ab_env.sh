#!/bin/bash
# GPU box: alternate bench runs with an environment switch on / off:  bash tools/ab_env.sh VAR [pairs]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; OUT=gpurun_out/ab_$1.txt; rm -f $OUT
for i in $(seq 1 ${2:-4}); do
  for v in 1 0; do
    env $1=$v python bench.py --steps 30 --warmup 5 --single-dtype --no-roofline --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1=$v', d['value'], d['outputs_finite'])" | tee -a $OUT
  done
done
