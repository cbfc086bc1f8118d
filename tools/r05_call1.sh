#!/bin/bash
# GPU box, round 5, first call: parity of the pipelined family, its A/B against the incumbents, stamps, a baseline bench line
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "pipelined" > gpurun_out/r05_c1_tests.log 2>&1
tail -5 gpurun_out/r05_c1_tests.log
timeout 600 python tools/gemm3_ab.py 3 > gpurun_out/r05_gemm3_ab.txt 2>gpurun_out/r05_gemm3_ab.err
cat gpurun_out/r05_gemm3_ab.txt
DD_TIMELINE_TILES=72,74,82 timeout 600 bash tools/gemm2_timeline.sh > gpurun_out/r05_gemm3_timeline.txt 2>gpurun_out/r05_gemm3_timeline.err
cat gpurun_out/r05_gemm3_timeline.txt
