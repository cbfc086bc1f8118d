#!/bin/bash
# Runs ON THE GPU BOX: gemm4 tests; the tracked tile table challenged with the pipelined tiles (their persistent form now
# exists); the whole GPU suite with the challenged table, saving every shape it tunes and minting the oracle caches; A/B.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gemm4_gpu.py -q -m gpu > gpurun_out/r06_gemm4_tests.log 2>&1
echo "gemm4 tests rc=$?" >> gpurun_out/r06_gemm4_tests.log
grep -E "passed|failed|FAILED" gpurun_out/r06_gemm4_tests.log | tail -12
T1=$PWD/gpurun_out/tuned_r06_challenged.json
cp dualdiff_amd/tuned/gfx950.json $T1
timeout 1500 python bench.py --challenge-tiles 72,73,75,78 --tune-cache $T1 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-extra-legs 2> gpurun_out/r06_challenge.err | tail -1 | cut -c1-300
grep "^\[tune\]" gpurun_out/r06_challenge.err > gpurun_out/r06_challenge.log; wc -l gpurun_out/r06_challenge.log
grep -c "->" gpurun_out/r06_challenge.log
T2=$PWD/gpurun_out/tuned_r06_suite.json
cp $T1 $T2
DD_TUNE_TABLE=$T1 DD_SAVE_TUNED=$T2 DD_MINT_ORACLE=$PWD/gpurun_out/oracle_cache timeout 1700 python -m pytest tests -q -m gpu --durations=25 > gpurun_out/r06_suite2.log 2>&1
echo "suite rc=$?" >> gpurun_out/r06_suite2.log
tail -40 gpurun_out/r06_suite2.log
ls -la gpurun_out/oracle_cache | tail -12
for i in 1 2 3; do
  for t in old new; do
    if [ $t = new ]; then export DD_TUNE_TABLE=$T2; else unset DD_TUNE_TABLE; fi
    python bench.py --steps 30 --warmup 5 --single-dtype --no-roofline --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('table=$t', round(d['value'],2), round(d['batched']['value'],2), d['outputs_finite'])" | tee -a gpurun_out/r06_table_ab.txt
  done
done
unset DD_TUNE_TABLE
for i in 1 2; do
  for t in on off; do
    if [ $t = off ]; then export DD_PERSIST3=0; else unset DD_PERSIST3; fi
    DD_TUNE_TABLE=$T2 python bench.py --steps 30 --warmup 5 --single-dtype --no-roofline --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new table, persist3=$t', round(d['value'],2), round(d['batched']['value'],2), d['env'])" | tee -a gpurun_out/r06_table_ab.txt
  done
done
