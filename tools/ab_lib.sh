#!/bin/bash
# GPU box: alternate bench runs between the in-tree library and another build of the same ABI:  bash tools/ab_lib.sh <other.so> [pairs]
# (one scene and four scenes per GPU; --allow-alt-lib: bench.py refuses DD_HIP_LIB without it since round 6)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; OUT=gpurun_out/ab_lib.txt; rm -f $OUT
for i in $(seq 1 ${2:-4}); do
  for v in new old; do
    if [ $v = old ]; then export DD_HIP_LIB=$GRAFT_REPO_ROOT/$1; else unset DD_HIP_LIB; fi
    python bench.py --steps 30 --warmup 5 --single-dtype --no-roofline --no-cpu-baseline --no-extra-legs --allow-alt-lib 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=$v', round(d['value'], 2), round((d.get('batched') or {}).get('value', 0), 2), d['outputs_finite'])" | tee -a $OUT
  done
done
