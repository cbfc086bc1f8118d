#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_properties_gpu.py -x -q -k "conv" > gpurun_out/r05_c23_tests.log 2>&1
tail -2 gpurun_out/r05_c23_tests.log
OUT=gpurun_out/r05_conv3s_prologue.txt; rm -f $OUT
AB_LABEL=new python3 tools/conv3s_ab.py 3 2>&1 | grep -v amdgpu.ids | tee -a $OUT
AB_LABEL=prev DD_HIP_LIB=$PWD/dualdiff_amd/lib/libdd_c3prev.so python3 tools/conv3s_ab.py 3 2>&1 | grep -v amdgpu.ids | tee -a $OUT
for i in 1 2 3; do
  for v in new prev; do
    if [ $v = prev ]; then export DD_HIP_LIB=$PWD/dualdiff_amd/lib/libdd_c3prev.so; else unset DD_HIP_LIB; fi
    timeout 300 python bench.py --steps 30 --warmup 5 --no-roofline --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prologue=$v', round(d['value'],2), round(d['other_dtype']['value'],2), round(d['batched']['value'],2), d['outputs_finite'])" | tee -a $OUT
  done
done
