#!/bin/bash
# GPU box: engine / memory clocks and power while the bench loop runs (what "peak" the step can physically see).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; OUT=gpurun_out/r06_clocks.txt
echo "# idle" > $OUT
rocm-smi --showclocks --showpower 2>&1 | grep -iE "sclk|mclk|fclk|power|socclk" | head -8 >> $OUT
python bench.py --steps 1500 --warmup 5 --single-dtype --no-roofline --no-cpu-baseline --no-extra-legs --batched-scenes 0 > gpurun_out/clock_bench.json 2>/dev/null &
BP=$!
sleep 45
for i in 1 2 3 4 5 6; do
  echo "# under load, sample $i" >> $OUT
  rocm-smi --showclocks --showpower 2>&1 | grep -iE "sclk|mclk|fclk|power|socclk" | head -8 >> $OUT
  sleep 1
done
wait $BP
python -c "import json; d=json.loads(open('gpurun_out/clock_bench.json').read().strip().splitlines()[-1]); print('# bench', d['value'], 'steps/s over', d['steps'], 'steps')" >> $OUT
cat $OUT
