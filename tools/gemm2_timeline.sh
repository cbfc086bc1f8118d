#!/bin/bash
# GPU box: builds the -DDD_DBG_STAMP diagnostic library (NOT the product) and prints the per-workgroup timeline of the
# dominant dense class's shapes:  bash tools/gemm2_timeline.sh > gpurun_out/r04_gemm2_timeline.txt
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
L=$R/dualdiff_amd/lib
python3 -c "import sys; sys.path.insert(0, '$R'); from dualdiff_amd import _build; _build.build_native()" 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -fPIC -Wno-unused-value -DNDEBUG -DDD_DBG_STAMP \
  -c $R/dualdiff_amd/csrc/gemm.hip -o /tmp/gemm_stamp.o
OBJS=$(ls $L/obj/*.o | grep -v "/gemm.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libdd_stamp.so /tmp/gemm_stamp.o $OBJS
cd $R
DD_HIP_LIB=/tmp/libdd_stamp.so DD_DBG_STAMP_WS=1 python3 tools/gemm2_timeline.py 2>&1 | grep -v amdgpu.ids
