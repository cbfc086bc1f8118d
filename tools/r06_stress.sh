#!/bin/bash
# GPU box: the persistent-walk tests repeated (a latent ordering race would show as an occasional mismatch), then the probe.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; OUT=gpurun_out/r06_stress.txt; rm -f $OUT
for i in $(seq 1 ${1:-12}); do
  timeout 600 python -m pytest tests/test_gemm4_gpu.py -q -m gpu 2>&1 | grep -E "^FAILED|passed|failed" | cut -c1-200 | tee -a $OUT
done
timeout 600 python tools/xs_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tee -a $OUT
