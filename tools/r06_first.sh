#!/bin/bash
# Runs ON THE GPU BOX: round-6 first contact — the forward-graph tests (varlen buckets, safe outputs), the whole GPU suite
# with its 40 slowest tests, and a bench line that also tunes the capacity-layout shapes into a side table.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_forward_graphs_gpu.py tests/test_dropin_loop_gpu.py -q -m gpu -x > gpurun_out/r06_fg_tests.log 2>&1
echo "fg tests rc=$?" >> gpurun_out/r06_fg_tests.log
tail -15 gpurun_out/r06_fg_tests.log
cp dualdiff_amd/tuned/gfx950.json gpurun_out/r06_table_in.json
timeout 900 python bench.py --steps 30 --warmup 5 --tune-cache gpurun_out/r06_table_out.json > gpurun_out/r06_bench0.json 2> gpurun_out/r06_bench0.err
echo "bench rc=$?"; cut -c1-1500 gpurun_out/r06_bench0.json; tail -5 gpurun_out/r06_bench0.err
timeout 1500 python -m pytest tests -q -m gpu --durations=60 > gpurun_out/r06_tests_durations.log 2>&1
echo "suite rc=$?" >> gpurun_out/r06_tests_durations.log
tail -75 gpurun_out/r06_tests_durations.log
