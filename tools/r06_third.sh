#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/gemm4_diag.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_gemm4_diag.txt | tail -60
timeout 1200 python -m pytest tests/test_gemm4_gpu.py tests/test_forward_graphs_gpu.py tests/test_batched_step_gpu.py -q -m gpu > gpurun_out/r06_gemm4_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r06_gemm4_tests.log
grep -E "passed|failed|FAILED|Error|rel-L2" gpurun_out/r06_gemm4_tests.log | tail -30
