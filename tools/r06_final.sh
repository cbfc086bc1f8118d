#!/bin/bash
# Runs ON THE GPU BOX: the whole GPU suite (parity CSV -> gpurun_out/r06_parity.csv), the sides table, the profile refresh.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/r06_parity.csv
timeout 1500 python -m pytest tests -q -m gpu --durations=12 > gpurun_out/r06_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r06_tests.log
tail -20 gpurun_out/r06_tests.log
timeout 1200 bash tools/gemm4_bound.sh > /dev/null 2>&1
grep -c . gpurun_out/r06_gemm3_bound.txt
bash tools/refresh_profiles.sh r06 > gpurun_out/r06_refresh.log 2>&1
tail -32 gpurun_out/r06_refresh.log
python bench.py --steps 30 --warmup 5 --fp8-weights --lora-rank 4 --no-cpu-baseline --no-roofline > gpurun_out/r06_bench_fp8.json 2>/dev/null
cut -c1-300 gpurun_out/r06_bench_fp8.json
