#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- 'bash tools/refresh_profiles.sh r01_v3'): bench line, rocprofv3
# kernel-trace stats of the same command, HBM-traffic PMC passes, timeline summary.
# Everything lands in gpurun_out/<tag>_*; copy what should be judged into profiles/.
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
T=$OUT/${TAG}_tune.json
mkdir -p $OUT
rm -f $T
cd /tmp && export TMPDIR=/tmp
# 1. the bench line (also writes the tile/split table the profiled runs reuse, so they time the
#    same kernels without the autotuner's trial launches)
python3 $R/bench.py --steps 30 --warmup 5 --tune-cache $T > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
# 2. kernel trace + stats of the same command
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/${TAG}_kt -o kt -- \
  python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --tune-cache $T \
  > $OUT/${TAG}_bench_under_rocprof.json 2> /dev/null
cp /tmp/${TAG}_kt/kt_kernel_stats.csv $OUT/${TAG}_kernel_stats.csv
python3 $R/tools/trace_summary.py /tmp/${TAG}_kt/kt_kernel_trace.csv 20 > $OUT/${TAG}_trace_summary.txt
# 3. HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes (eager launches, one stream)
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/${TAG}_$C -o p -- \
    python3 $R/bench.py --steps 2 --warmup 1 --no-graph --serial-branches --no-cpu-baseline --no-roofline \
    --tune-cache $T > /dev/null 2>&1
done
python3 $R/tools/pmc_summary.py /tmp/${TAG}_FETCH_SIZE /tmp/${TAG}_WRITE_SIZE $OUT/${TAG}_pmc_traffic.json $OUT/${TAG}_pmc_traffic.csv
# 3b. MFMA-pipe and VALU-issue utilisation per kernel (one more counter pass, same eager command)
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU \
  --output-format csv -d /tmp/${TAG}_MFMA -o p -- \
  python3 $R/bench.py --steps 2 --warmup 1 --no-graph --serial-branches --no-cpu-baseline --no-roofline \
  --tune-cache $T > /dev/null 2>&1
python3 $R/tools/pmc_mfma_summary.py /tmp/${TAG}_MFMA $OUT/${TAG}_pmc_mfma.csv
# 4. per-shape table of one eager step
DD_TUNE_CACHE=$T python3 $R/tools/step_shapes.py > $OUT/${TAG}_step_shapes.txt 2>&1
cat $OUT/${TAG}_bench.json
cat $OUT/${TAG}_trace_summary.txt
