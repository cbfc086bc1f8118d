#!/bin/bash
# GPU box: the direct / band 3x3 conv against its own sides: product build and four diagnostic builds of
# csrc/gemm.hip (matrix instructions / activation gathers / weight-fragment reads / loop barrier removed), timed hot.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD; L=$R/dualdiff_amd/lib
[ -f $L/obj/norm.o ] || python3 -c "from dualdiff_amd import _build; _build.build_native(force=True)" 2>/dev/null
VARS="${C3_VARIANTS:-EARLYDMA NOMFMA NOGATHER NOWREAD NOBAR}"
for V in $VARS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -fPIC -Wno-unused-value -DNDEBUG -mllvm -amdgpu-mfma-vgpr-form=1 \
    -DDD_DBG_ONLY_C3 $(for f in $(echo $V | tr + ' '); do echo -n "-DDD_DBG_C3_$f "; done) -c $R/dualdiff_amd/csrc/gemm.hip -o /tmp/gemm_$V.o &
done
wait
OBJS=$(ls $L/obj/*.o | grep -v "/gemm.o")
OUT=gpurun_out/${C3_OUT:-r05_conv3s_bound.txt}; rm -f $OUT
python3 tools/conv3s_sides.py product 2>&1 | grep -v amdgpu.ids | tee -a $OUT
for V in $VARS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libdd_c3_$V.so /tmp/gemm_$V.o $OBJS
  DD_HIP_LIB=/tmp/libdd_c3_$V.so python3 tools/conv3s_sides.py $V 2>&1 | grep -v amdgpu.ids | tee -a $OUT
done
