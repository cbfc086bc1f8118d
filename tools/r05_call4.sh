#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_forward_graphs_gpu.py tests/test_dropin_loop_gpu.py -x -q > gpurun_out/r05_c4_tests.log 2>&1
tail -5 gpurun_out/r05_c4_tests.log
timeout 600 python tools/dropin_breakdown.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_dropin_breakdown.txt
timeout 600 python bench.py --serial-branches --steps 30 --warmup 5 --no-roofline --no-cpu-baseline --single-dtype --batched-scenes 0 --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused, ONE stream:', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r05_dropin_breakdown.txt
