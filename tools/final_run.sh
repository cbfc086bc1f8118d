#!/bin/bash
# Runs ON THE GPU BOX: the whole GPU suite (parity CSV -> gpurun_out/<tag>_parity.csv) followed by the profile refresh.
TAG=${1:-r04}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/r05_parity.csv
timeout 4200 python -m pytest tests -q -m gpu -x > gpurun_out/${TAG}_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/${TAG}_tests.log
tail -4 gpurun_out/${TAG}_tests.log
bash tools/refresh_profiles.sh $TAG > gpurun_out/${TAG}_refresh.log 2>&1
python bench.py --steps 30 --warmup 5 --fp8-weights --lora-rank 4 --no-cpu-baseline --no-roofline > gpurun_out/${TAG}_bench_fp8.json 2>/dev/null
cut -c1-400 gpurun_out/${TAG}_bench.json; cat gpurun_out/${TAG}_trace_summary.txt | head -14
