#!/bin/bash
cd $GRAFT_REPO_ROOT
G4_VARIANTS="NOSTORE NOLDS NOSTORE+NOLDS" G4_OUT=r06_gemm4_extra.txt timeout 900 bash tools/gemm4_bound.sh > /dev/null 2>&1
grep -v "PERSIST3=0" gpurun_out/r06_gemm4_extra.txt | cut -c1-330
