#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gemm4_gpu.py -q -m gpu -x > gpurun_out/r06_gemm4_tests.log 2>&1
echo "gemm4 tests rc=$?" >> gpurun_out/r06_gemm4_tests.log
grep -E "passed|failed|FAILED" gpurun_out/r06_gemm4_tests.log | tail -8
timeout 600 python tools/gemm4_sides.py product 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_gemm4_roles.txt
DD_PERSIST3=0 timeout 600 python tools/gemm4_sides.py "PERSIST3=0" 2>&1 | grep -v amdgpu.ids | grep -v "^ln-out" | tee -a gpurun_out/r06_gemm4_roles.txt
