#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "band_direct or small_image_direct" > gpurun_out/r05_c11_tests.log 2>&1
tail -4 gpurun_out/r05_c11_tests.log
timeout 900 python tools/conv3s_ab.py 3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_conv3s_ab.txt
