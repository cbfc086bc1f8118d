#!/bin/bash
# GPU box: the attention kernel against its own ceilings: product build, exponentials replaced by moves
# (-DDD_DBG_NOEXP), matrix instructions removed (-DDD_DBG_NOMFMA), K/V staging removed (-DDD_DBG_NOSTAGE)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD; L=$R/dualdiff_amd/lib
[ -f $L/obj/norm.o ] || python3 -c "from dualdiff_amd import _build; _build.build_native(force=True)" 2>/dev/null
for V in NOMFMA NOEXP NOSTAGE; do
  v=$(echo $V | tr A-Z a-z)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -fPIC -Wno-unused-value -DNDEBUG -mllvm -amdgpu-mfma-vgpr-form=1 \
    -fno-honor-nans -DDD_DBG_$V -c $R/dualdiff_amd/csrc/attention.hip -o /tmp/attn_$v.o &
done
wait
OBJS=$(ls $L/obj/*.o | grep -v "/attention.o")
OUT=gpurun_out/r05_attn_bound.txt; rm -f $OUT
python3 tools/attn_sides.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT
for v in noexp nomfma nostage; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libdd_attn_$v.so /tmp/attn_$v.o $OBJS
  DD_HIP_LIB=/tmp/libdd_attn_$v.so python3 tools/attn_sides.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT
done
