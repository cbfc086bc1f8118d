#!/bin/bash
# GPU box: direct conv with the fragments gathered inside the MFMA block (single fragment buffer, compile-time wave
# roles, disjoint DMA windows) against the library of the previous commit (dualdiff_amd/lib/libdd_c3old.so, prebuilt)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_properties_gpu.py -x -q -k "conv" > gpurun_out/r05_c17_tests.log 2>&1
tail -3 gpurun_out/r05_c17_tests.log
OLD=$PWD/dualdiff_amd/lib/libdd_c3old.so
OUT=gpurun_out/r05_conv3s_fine.txt; rm -f $OUT
python3 tools/conv3s_sides.py new 2>&1 | grep -v amdgpu.ids | tee -a $OUT
DD_HIP_LIB=$OLD python3 tools/conv3s_sides.py old 2>&1 | grep -v amdgpu.ids | tee -a $OUT
for i in 1 2 3; do
  for v in new old; do
    if [ $v = old ]; then export DD_HIP_LIB=$OLD; else unset DD_HIP_LIB; fi
    timeout 300 python bench.py --steps 30 --warmup 5 --no-roofline --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('conv step=$v', round(d['value'],2), round(d['other_dtype']['value'],2), round(d['batched']['value'],2), d['outputs_finite'])" | tee -a $OUT
  done
done
