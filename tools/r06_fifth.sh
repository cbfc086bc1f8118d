#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gemm4_gpu.py -q -m gpu > gpurun_out/r06_gemm4_tests.log 2>&1
echo "gemm4 tests rc=$?" >> gpurun_out/r06_gemm4_tests.log
grep -E "passed|failed|FAILED" gpurun_out/r06_gemm4_tests.log | tail -8
rm -f gpurun_out/r06_parity.csv
timeout 1500 python -m pytest tests -q -m gpu --durations=25 > gpurun_out/r06_suite3.log 2>&1
echo "suite rc=$?" >> gpurun_out/r06_suite3.log
tail -34 gpurun_out/r06_suite3.log
timeout 1200 bash tools/gemm4_bound.sh > /dev/null 2>&1
cut -c1-400 gpurun_out/r06_gemm3_bound.txt
bash tools/ab_lib.sh dualdiff_amd/lib/libdd_prev.so 3
bash tools/refresh_profiles.sh r06 > gpurun_out/r06_refresh.log 2>&1
tail -30 gpurun_out/r06_refresh.log
