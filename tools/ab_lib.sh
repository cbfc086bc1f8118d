#!/bin/bash
# GPU box: alternate bench runs between the in-tree library and another build of the same ABI:  bash tools/ab_lib.sh <other.so> [pairs]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; OUT=gpurun_out/ab_lib.txt; rm -f $OUT
for i in $(seq 1 ${2:-4}); do
  for v in new old; do
    if [ $v = old ]; then export DD_HIP_LIB=$GRAFT_REPO_ROOT/$1; else unset DD_HIP_LIB; fi
    python bench.py --steps 30 --warmup 5 --single-dtype --no-roofline --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=$v', d['value'], d['outputs_finite'])" | tee -a $OUT
  done
done
