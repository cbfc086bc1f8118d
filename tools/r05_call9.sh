#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out; export TMPDIR=/tmp
for m in 0 1 2 4 8 16 24; do
  echo "== DD_FWD_FORK=$m" | tee -a gpurun_out/r05_dropin_forks.txt
  DD_FWD_FORK=$m timeout 600 python tools/dropin_breakdown.py 2>&1 | grep -v amdgpu.ids | grep "forward graph\|whole loop" | tee -a gpurun_out/r05_dropin_forks.txt
done
