#!/bin/bash
# Diagnostic builds of the C-ABI library (NOT the product): libdd_nomfma.so drops the matrix instructions of the fp16
# GEMM main loops, libdd_nodma.so the LDS-DMA loads; load one with DD_HIP_LIB=... to see which side bounds a kernel.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
L=$R/dualdiff_amd/lib
python3 -c "import sys; sys.path.insert(0, '$R'); from dualdiff_amd import _build; _build.build_native()"
for V in NOMFMA NODMA; do
  v=$(echo $V | tr A-Z a-z)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DNDEBUG -DDD_DBG_$V \
    -c $R/dualdiff_amd/csrc/gemm.hip -o /tmp/gemm_$v.o &
done
wait
for v in nomfma nodma; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libdd_$v.so /tmp/gemm_$v.o $L/obj/norm.o $L/obj/attention.o $L/obj/elementwise.o
done
# attention: matrix instructions removed / exponentials replaced by moves (tools/attn_sides.py)
for V in NOMFMA NOEXP NOSTAGE; do
  v=$(echo $V | tr A-Z a-z)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DNDEBUG -mllvm -amdgpu-mfma-vgpr-form=1 \
    -fno-honor-nans -DDD_DBG_$V -c $R/dualdiff_amd/csrc/attention.hip -o /tmp/attn_$v.o &
done
wait
for v in nomfma noexp nostage; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libdd_attn_$v.so $L/obj/gemm.o $L/obj/norm.o /tmp/attn_$v.o $L/obj/elementwise.o
done
ls -la $L/*.so
