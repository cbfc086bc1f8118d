#!/bin/bash
# Runs ON THE GPU BOX (CPU only): mints the oracle DDIM-50 trajectory parts into gpurun_out/mint/
# (tests/golden/mint_trajectory.py), two modes in parallel.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out/mint
export DD_MINT_OUT=$R/gpurun_out/mint
N=$(nproc); T=$(( N / 2 > 24 ? 24 : N / 2 ))
echo "cores $N threads/mode $T"
python3 $R/tests/golden/mint_trajectory.py ref $T > $R/gpurun_out/mint/ref.log 2>&1 &
P1=$!
python3 $R/tests/golden/mint_trajectory.py floor_f16 $T > $R/gpurun_out/mint/floor_f16.log 2>&1 &
P2=$!
wait $P1 $P2
python3 $R/tests/golden/mint_trajectory.py floor_bf16 $(( T * 2 )) 3 > $R/gpurun_out/mint/floor_bf16.log 2>&1
tail -n 3 $R/gpurun_out/mint/*.log
