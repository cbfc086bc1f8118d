#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; OUT=gpurun_out/r06_nop.txt; rm -f $OUT
L=$GRAFT_REPO_ROOT/dualdiff_amd/lib
echo "--- xs_cols patch + s_nop 4 (libdd_xsnop.so)" | tee -a $OUT
for i in 1 2 3; do DD_HIP_LIB=$L/libdd_xsnop.so timeout 600 python -m pytest tests/test_gemm4_gpu.py -q -m gpu 2>&1 | grep -E "^FAILED|passed|failed" | cut -c1-200 | tee -a $OUT; done
echo "--- product (s_nop 4, no xs_cols)" | tee -a $OUT
for i in 1 2 3; do timeout 600 python -m pytest tests/test_gemm4_gpu.py -q -m gpu 2>&1 | grep -E "^FAILED|passed|failed" | cut -c1-200 | tee -a $OUT; done
bash tools/ab_lib.sh dualdiff_amd/lib/libdd_prev.so 3 2>&1 | tee -a $OUT
