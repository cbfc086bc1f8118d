#!/bin/bash
# GPU box: the direct conv's per-step DMAs moved out of the barrier's shadow (issued under the other wave's MFMAs):
# conv tests, the kernel against its own sides, and a whole-step A/B against the old order (-DDD_DBG_C3_EARLYDMA).
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD; L=$R/dualdiff_amd/lib
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_properties_gpu.py -x -q -k "conv" > gpurun_out/r05_c14_tests.log 2>&1
tail -3 gpurun_out/r05_c14_tests.log
[ -f $L/obj/norm.o ] || python3 -c "from dualdiff_amd import _build; _build.build_native(force=True)" 2>/dev/null
# the old order as a full library (same ABI) for the step A/B, built while the kernel-level runs go
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -fPIC -Wno-unused-value -DNDEBUG -mllvm -amdgpu-mfma-vgpr-form=1 \
  -DDD_DBG_C3_EARLYDMA -c $R/dualdiff_amd/csrc/gemm.hip -o /tmp/gemm_earlydma_full.o &
C3_VARIANTS="EARLYDMA NOMFMA NOGATHER NOWREAD NOBAR NOMFMA+NOBAR" bash tools/conv3s_bound.sh
wait
OBJS=$(ls $L/obj/*.o | grep -v "/gemm.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libdd_earlydma.so /tmp/gemm_earlydma_full.o $OBJS
OUT=gpurun_out/r05_c3dma_ab.txt; rm -f $OUT
for i in 1 2 3; do
  for v in new old; do
    if [ $v = old ]; then export DD_HIP_LIB=/tmp/libdd_earlydma.so; else unset DD_HIP_LIB; fi
    timeout 300 python bench.py --steps 30 --warmup 5 --no-roofline --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('conv dma order=$v', round(d['value'],2), round(d['other_dtype']['value'],2), round(d['batched']['value'],2), d['outputs_finite'])" | tee -a $OUT
  done
done
