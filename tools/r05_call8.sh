#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
bash tools/refresh_profiles.sh r05 > gpurun_out/r05_refresh.log 2>&1
tail -30 gpurun_out/r05_refresh.log
