#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1700 python -m pytest tests/test_resolutions_gpu.py tests/test_tokens_gpu.py -q -k "resolution or bounds" > gpurun_out/r05_c7_tests.log 2>&1
tail -40 gpurun_out/r05_c7_tests.log
