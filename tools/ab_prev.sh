#!/bin/bash
# GPU box: a test selection, then alternating bench runs of the in-tree library against dualdiff_amd/lib/libdd_prev.so (the
# previous build, copied there before the rebuild):  bash tools/ab_prev.sh "<pytest -k expression>" <out tag> [rounds] [pre-command]
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out; export TMPDIR=/tmp
K="$1"; TAG=$2; N=${3:-3}
if [ -n "$K" ]; then
  timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_properties_gpu.py tests/test_golden_direct_gpu.py tests/test_fp8_mfma_gpu.py -x -q -k "$K" > gpurun_out/${TAG}_tests.log 2>&1
  tail -2 gpurun_out/${TAG}_tests.log
fi
[ -n "$4" ] && eval "$4"
OUT=gpurun_out/${TAG}_ab.txt; rm -f $OUT
for i in $(seq 1 $N); do
  for v in new prev; do
    if [ $v = prev ]; then export DD_HIP_LIB=$PWD/dualdiff_amd/lib/libdd_prev.so; else unset DD_HIP_LIB; fi
    timeout 300 python bench.py --steps 30 --warmup 5 --no-roofline --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=$v', round(d['value'],2), round(d['other_dtype']['value'],2), round(d['batched']['value'],2), d['outputs_finite'])" | tee -a $OUT
  done
done
