#!/bin/bash
# GPU box: the GEGLU GEMMs against their own sides (full builds of gemm.hip with -DDD_DBG_NODMA / -DDD_DBG_NOMFMA)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD; L=$R/dualdiff_amd/lib
[ -f $L/obj/norm.o ] || python3 -c "from dualdiff_amd import _build; _build.build_native(force=True)" 2>/dev/null
VARS="NODMA NOMFMA NODMA+NOMFMA"
for V in $VARS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -fPIC -Wno-unused-value -DNDEBUG -mllvm -amdgpu-mfma-vgpr-form=1 \
    $(for f in $(echo $V | tr + ' '); do echo -n "-DDD_DBG_$f "; done) -c $R/dualdiff_amd/csrc/gemm.hip -o /tmp/gemm_$V.o &
done
wait
OBJS=$(ls $L/obj/*.o | grep -v "/gemm.o")
OUT=gpurun_out/r05_geglu_bound.txt; rm -f $OUT
python3 tools/geglu_sides.py product 2>&1 | grep -v amdgpu.ids | tee -a $OUT
for V in $VARS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libdd_gg_$V.so /tmp/gemm_$V.o $OBJS
  DD_HIP_LIB=/tmp/libdd_gg_$V.so python3 tools/geglu_sides.py $V 2>&1 | grep -v amdgpu.ids | tee -a $OUT
done
