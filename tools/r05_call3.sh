#!/bin/bash
# GPU box, round 5: parity of the pipelined family, A/B, then challenge the tracked tile table with it and A/B the tables
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "pipelined" > gpurun_out/r05_c3_tests.log 2>&1
tail -3 gpurun_out/r05_c3_tests.log
timeout 600 python tools/gemm3_ab.py 3 > gpurun_out/r05_gemm3_ab.txt 2>gpurun_out/r05_gemm3_ab.err
cat gpurun_out/r05_gemm3_ab.txt
NEW=gpurun_out/tuned_r05.json
cp dualdiff_amd/tuned/gfx950.json $NEW
timeout 1200 python bench.py --challenge-tiles ${1:-72,73,74,75,76,77,78,79} --tune-cache $NEW --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>gpurun_out/r05_challenge.err | grep "^\[tune\]" > gpurun_out/r05_challenge.log
wc -l gpurun_out/r05_challenge.log
for i in 1 2 3; do
  for t in old new; do
    if [ $t = new ]; then export DD_TUNE_TABLE=$PWD/$NEW; else unset DD_TUNE_TABLE; fi
    timeout 300 python bench.py --steps 30 --warmup 5 --no-roofline --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('table=$t', round(d['value'],2), round(d['other_dtype']['value'],2), round(d['batched']['value'],2))" | tee -a gpurun_out/r05_table_ab.txt
  done
done
