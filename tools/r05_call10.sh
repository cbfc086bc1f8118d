#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/r05_vgprform_ab.txt; rm -f $OUT
for i in 1 2 3; do
  for v in base vgprform; do
    if [ $v = vgprform ]; then export DD_HIP_LIB=$PWD/dualdiff_amd/lib/libdd_vgprform.so; else unset DD_HIP_LIB; fi
    timeout 300 python bench.py --steps 30 --warmup 5 --no-roofline --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=$v', round(d['value'],2), round(d['other_dtype']['value'],2), round(d['batched']['value'],2), d['outputs_finite'])" | tee -a $OUT
  done
done
