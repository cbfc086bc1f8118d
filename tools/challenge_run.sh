#!/bin/bash
# GPU box: challenge the tracked tile table with new tiles ($1, e.g. 52), then A/B old vs new table on the same box.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
NEW=gpurun_out/tuned_challenge.json
cp dualdiff_amd/tuned/gfx950.json $NEW
python bench.py --challenge-tiles $1 --tune-cache $NEW --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | grep "^\[tune\]" > gpurun_out/r03h_challenge.log
wc -l gpurun_out/r03h_challenge.log
for i in 1 2 3 4; do
  for t in old new; do
    if [ $t = new ]; then export DD_TUNE_TABLE=$GRAFT_REPO_ROOT/$NEW; else unset DD_TUNE_TABLE; fi
    python bench.py --steps 30 --warmup 5 --single-dtype --no-roofline --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('table=$t', d['value'])" | tee -a gpurun_out/r03h_table_ab.txt
  done
done
