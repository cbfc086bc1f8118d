#!/bin/bash
# store-policy / store-count experiments on the all-wave persistent kernel
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD; L=$R/dualdiff_amd/lib
for V in STORE_AUX=2 STORE_AUX=16 STORE_AUX=18 ONESTORE; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -fPIC -Wno-unused-value -DNDEBUG -mllvm -amdgpu-mfma-vgpr-form=1 -DDD_DBG_$V -c $R/dualdiff_amd/csrc/gemm.hip -o /tmp/gemm_$V.o &
done
wait
OBJS=$(ls $L/obj/*.o | grep -v "/gemm.o")
OUT=gpurun_out/r06_store_policy.txt; rm -f $OUT
python3 tools/gemm4_sides.py product 2>&1 | grep -v amdgpu.ids | grep -v "^ln-out" | tee -a $OUT | cut -c1-420
for V in STORE_AUX=2 STORE_AUX=16 STORE_AUX=18 ONESTORE; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libdd_$V.so /tmp/gemm_$V.o $OBJS
  DD_HIP_LIB=/tmp/libdd_$V.so python3 tools/gemm4_sides.py $V 2>&1 | grep -v amdgpu.ids | grep -v "^ln-out" | tee -a $OUT | cut -c1-420
done
