#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gemm4_gpu.py -q -m gpu -x 2>&1 | tail -2
python3 tools/gemm4_sides.py sector 2>&1 | grep -v amdgpu.ids | grep -v "^ln-out" | tee gpurun_out/r06_sector.txt | cut -c1-600
DD_HIP_LIB=$PWD/dualdiff_amd/lib/libdd_prev.so python3 tools/gemm4_sides.py interleaved 2>&1 | grep -v amdgpu.ids | grep -v "^ln-out" | tee -a gpurun_out/r06_sector.txt | cut -c1-600
