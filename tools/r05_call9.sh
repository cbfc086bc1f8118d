#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_forward_graphs_gpu.py tests/test_dropin_loop_gpu.py -x -q > gpurun_out/r05_c9_tests.log 2>&1
tail -15 gpurun_out/r05_c9_tests.log
timeout 600 python tools/dropin_breakdown.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_dropin_breakdown2.txt
timeout 600 python bench.py --steps 30 --warmup 5 --no-roofline --no-cpu-baseline --single-dtype --batched-scenes 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused:', d['value'], 'dropin:', d.get('dropin'), 'unipc20:', d.get('unipc20'))" | tee -a gpurun_out/r05_dropin_breakdown2.txt
