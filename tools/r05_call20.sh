#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash tools/r05_call17.sh
C3_OUT=r05_conv3s_bound4.txt C3_VARIANTS="NOGATHER NOWREAD NOBAR NODMA NOMFMA" bash tools/conv3s_bound.sh
