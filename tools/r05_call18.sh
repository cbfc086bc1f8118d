#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash tools/r05_call17.sh
C3_OUT=r05_conv3s_segments3.txt bash tools/r05_call16.sh 2>&1 | grep -v "tables\|prologue\|first 9\|start spread"
