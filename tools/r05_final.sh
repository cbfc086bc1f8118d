#!/bin/bash
# GPU box: whole GPU suite + profile refresh (tools/final_run.sh r05 without the fp8 leg first)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/r05_parity.csv
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r05_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r05_tests.log
tail -4 gpurun_out/r05_tests.log
bash tools/refresh_profiles.sh r05 > gpurun_out/r05_refresh.log 2>&1
tail -28 gpurun_out/r05_refresh.log
timeout 600 python bench.py --steps 30 --warmup 5 --fp8-weights --lora-rank 4 --no-cpu-baseline --no-roofline --no-extra-legs > gpurun_out/r05_bench_fp8.json 2>/dev/null
cut -c1-300 gpurun_out/r05_bench_fp8.json
