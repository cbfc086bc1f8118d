#!/bin/bash
# Mints tests/golden/oracle_cache/*.npz: runs the tests that cache their oracle outputs (tests/parity_util.oracle_cache)
# with DD_MINT_ORACLE set, i.e. every cached case is recomputed by the live CPU oracle, written, and compared with the
# stored copy.  The tests are GPU tests (the HIP path is compared in the same run): run on a GPU box,
#   gpurun -- 'bash tests/golden/mint_oracle_cache.sh'    then   cp gpurun_out/oracle_cache/*.npz tests/golden/oracle_cache/
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
mkdir -p gpurun_out/oracle_cache
DD_MINT_ORACLE=$PWD/gpurun_out/oracle_cache python -m pytest -q -m gpu \
  tests/test_parity_r03_gpu.py::test_full_step_bench_context tests/test_parity_r03_gpu.py::test_video_unet_forward_T8 \
  tests/test_video_gpu.py::test_video_unet_forward tests/test_model_gpu.py::test_unet_multiview_forward \
  tests/test_resolutions_gpu.py::test_unet_forward_other_resolution
ls -la gpurun_out/oracle_cache
