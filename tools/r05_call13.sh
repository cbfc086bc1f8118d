#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "pipelined or layernorm_emitting" > gpurun_out/r05_c13_tests.log 2>&1
tail -4 gpurun_out/r05_c13_tests.log
OUT=gpurun_out/r05_lnout_ab.txt; rm -f $OUT
for i in 1 2 3; do
  for v in 74 40; do
    DD_LN_OUT_TILE=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-roofline --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ln_out tile=$v', round(d['value'],2), round(d['other_dtype']['value'],2), round(d['batched']['value'],2), d['outputs_finite'])" | tee -a $OUT
  done
done
