#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_forward_graphs_gpu.py tests/test_dropin_loop_gpu.py -x -q > gpurun_out/r05_c4_tests.log 2>&1
tail -25 gpurun_out/r05_c4_tests.log
timeout 900 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r05_c4_bench.json 2>gpurun_out/r05_c4_bench.err
tail -5 gpurun_out/r05_c4_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_c4_bench.json').read().strip().splitlines()[-1])
print(len(json.dumps(d)), 'bytes')
for k in ('value','other_dtype','batched','unipc20','dropin'):
    print(k, json.dumps(d.get(k))[:600])
print('roofline', json.dumps(d['roofline'])[:900])
PY
