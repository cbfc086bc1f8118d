#!/bin/bash
# GPU box: the pipelined dense family (one tile per workgroup: dd_gemm3, persistent walk: dd_gemm4) against its own sides:
# product, product with DD_PERSIST3=0, no LDS-DMA (-DDD_DBG_NODMA), no matrix instructions (-DDD_DBG_NOMFMA), neither.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD; L=$R/dualdiff_amd/lib
[ -f $L/obj/norm.o ] || python3 -c "from dualdiff_amd import _build; _build.build_native(force=True)" 2>/dev/null
VARS="${G4_VARIANTS:-NODMA NOMFMA NODMA+NOMFMA NODMA+NOMFMA+NOSTORE NODMA+NOMFMA+NOSTORE+NOLDS}"
for V in $VARS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -fPIC -Wno-unused-value -DNDEBUG -mllvm -amdgpu-mfma-vgpr-form=1 \
    $(for f in $(echo $V | tr + ' '); do echo -n "-DDD_DBG_$f "; done) -c $R/dualdiff_amd/csrc/gemm.hip -o /tmp/gemm_$V.o &
done
wait
OBJS=$(ls $L/obj/*.o | grep -v "/gemm.o")
OUT=gpurun_out/${G4_OUT:-r06_gemm3_bound.txt}; rm -f $OUT
python3 tools/gemm4_sides.py product 2>&1 | grep -v amdgpu.ids | tee -a $OUT
DD_PERSIST3=0 python3 tools/gemm4_sides.py "product PERSIST3=0" 2>&1 | grep -v amdgpu.ids | grep -v "^ln-out" | tee -a $OUT
for V in $VARS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libdd_g4_$V.so /tmp/gemm_$V.o $OBJS
  DD_HIP_LIB=/tmp/libdd_g4_$V.so python3 tools/gemm4_sides.py $V 2>&1 | grep -v amdgpu.ids | tee -a $OUT
  DD_PERSIST3=0 DD_HIP_LIB=/tmp/libdd_g4_$V.so python3 tools/gemm4_sides.py "$V PERSIST3=0" 2>&1 | grep -v amdgpu.ids | tee -a $OUT
done
