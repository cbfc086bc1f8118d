#!/bin/bash
# Runs ON THE GPU BOX: correctness of the persistent pipelined GEMM (bit-identical to the dd_gemm2 twins), the GEGLU /
# feed-forward op tests (new gate function), then the sides table.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gemm4_gpu.py -q -m gpu -x > gpurun_out/r06_gemm4_tests.log 2>&1
echo "gemm4 tests rc=$?" >> gpurun_out/r06_gemm4_tests.log
tail -25 gpurun_out/r06_gemm4_tests.log
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "geglu or gemm or ff or feed" > gpurun_out/r06_ops_tests.log 2>&1
echo "ops tests rc=$?" >> gpurun_out/r06_ops_tests.log
tail -5 gpurun_out/r06_ops_tests.log
timeout 1500 bash tools/gemm4_bound.sh 2>&1 | tail -40
