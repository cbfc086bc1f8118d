#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- 'bash tools/refresh_profiles.sh r02'): bench line, rocprofv3 kernel-trace stats of
# the same command, HBM-traffic PMC passes, MFMA utilisation pass, timeline summary, per-shape table.
# Every run uses the TRACKED tile table (dualdiff_amd/tuned/gfx950.json): same kernels in every process.
# Everything lands in gpurun_out/<tag>_*; copy what should be judged into profiles/.
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. the bench line (default command of the driver + the per-class table)
DD_BENCH_KERNEL_TABLE=$OUT/${TAG}_classes.txt python3 $R/bench.py --steps 30 --warmup 5 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
# 2. kernel trace + stats of the same command (graph replay, 3 streams)
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/${TAG}_kt -o kt -- \
  python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --single-dtype --batched-scenes 0 --no-extra-legs \
  > $OUT/${TAG}_bench_under_rocprof.json 2> /dev/null
cp /tmp/${TAG}_kt/kt_kernel_stats.csv $OUT/${TAG}_kernel_stats.csv
python3 $R/tools/trace_summary.py /tmp/${TAG}_kt/kt_kernel_trace.csv 20 > $OUT/${TAG}_trace_summary.txt
# 2b. kernel stats of the INSTRUMENTED command the roofline line is computed from (eager, one stream): the
#     average duration of the dominant kernel in this file must agree with roofline.avg_us
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/${TAG}_ke -o ke -- \
  python3 $R/bench.py --steps 3 --warmup 1 --no-graph --serial-branches --no-cpu-baseline --no-roofline --single-dtype --batched-scenes 0 --no-extra-legs \
  > /dev/null 2>&1
cp /tmp/${TAG}_ke/ke_kernel_stats.csv $OUT/${TAG}_kernel_stats_eager.csv
# 3. HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes (eager launches, one stream)
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/${TAG}_$C -o p -- \
    python3 $R/bench.py --steps 2 --warmup 1 --no-graph --serial-branches --no-cpu-baseline --no-roofline --single-dtype --batched-scenes 0 --no-extra-legs \
    > /dev/null 2>&1
done
python3 $R/tools/pmc_summary.py /tmp/${TAG}_FETCH_SIZE /tmp/${TAG}_WRITE_SIZE $OUT/${TAG}_pmc_traffic.json $OUT/${TAG}_pmc_traffic.csv
# 3b. MFMA-pipe and VALU-issue utilisation per kernel (one more counter pass, same eager command)
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU \
  --output-format csv -d /tmp/${TAG}_MFMA -o p -- \
  python3 $R/bench.py --steps 2 --warmup 1 --no-graph --serial-branches --no-cpu-baseline --no-roofline --single-dtype --batched-scenes 0 --no-extra-legs \
  > /dev/null 2>&1
python3 $R/tools/pmc_mfma_summary.py /tmp/${TAG}_MFMA $OUT/${TAG}_pmc_mfma.csv
# 4. per-shape table of one eager step
python3 $R/tools/step_shapes.py > $OUT/${TAG}_step_shapes.txt 2>&1
cat $OUT/${TAG}_bench.json | cut -c1-600
cat $OUT/${TAG}_trace_summary.txt
# 5. BASELINE configs[2]: the SFA module on its own (3-launch rows + the fused kernel, 48 and 12 instances)
python3 $R/tools/sfa_roofline.py fp16 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_sfa_roofline.txt
