#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm4_gpu.py tests/test_ops_gpu.py -q -m gpu -x -k "gemm or geglu or persistent" 2>&1 | tail -2
rm -f gpurun_out/r06_xs_cols.txt
for C in 1 0 2 4 8; do DD_XS_COLS=$C python tools/xs_cols_ab.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06_xs_cols.txt | cut -c1-700; done
