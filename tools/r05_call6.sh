#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD; L=$R/dualdiff_amd/lib
[ -f $L/obj/norm.o ] || python3 -c "from dualdiff_amd import _build; _build.build_native(force=True)" 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -fPIC -Wno-unused-value -DNDEBUG -DDD_DBG_STAMP -c $R/dualdiff_amd/csrc/gemm.hip -o /tmp/gemm_stamp.o
OBJS=$(ls $L/obj/*.o | grep -v "/gemm.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libdd_stamp.so /tmp/gemm_stamp.o $OBJS
DD_HIP_LIB=/tmp/libdd_stamp.so DD_DBG_STAMP_WS=1 timeout 600 python3 tools/conv3s_stamps.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_conv3s_stamps.txt
