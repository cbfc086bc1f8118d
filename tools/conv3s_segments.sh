#!/bin/bash
# GPU box: per-segment clocks of the direct conv's steady-state step (stamp build)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD; L=$R/dualdiff_amd/lib
[ -f $L/obj/norm.o ] || python3 -c "from dualdiff_amd import _build; _build.build_native(force=True)" 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -fPIC -Wno-unused-value -DNDEBUG -mllvm -amdgpu-mfma-vgpr-form=1 \
  -DDD_DBG_ONLY_C3 -DDD_DBG_STAMP ${C3_EXTRA} -c $R/dualdiff_amd/csrc/gemm.hip -o /tmp/gemm_stamp.o
OBJS=$(ls $L/obj/*.o | grep -v "/gemm.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libdd_stamp.so /tmp/gemm_stamp.o $OBJS
DD_DBG_STAMP_WS=1 DD_HIP_LIB=/tmp/libdd_stamp.so python3 tools/conv3s_stamps.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${C3_OUT:-r05_conv3s_segments.txt}
