#!/bin/bash
# GPU box: sample rocm-smi clocks / power while the captured step replays, fp16 then bf16.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; OUT=gpurun_out/r03m_clocks.txt; rm -f $OUT
for dt in fp16 bf16; do
  echo "== $dt" >> $OUT
  python bench.py --steps 1500 --warmup 20 --no-roofline --no-cpu-baseline --single-dtype --dtype $dt > /tmp/b_$dt.json 2>/dev/null &
  PID=$!
  sleep 9
  for i in 1 2 3 4 5; do
    /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|fclk" | tr -s ' ' | head -6 >> $OUT
    echo "--" >> $OUT
    sleep 2
  done
  wait $PID
  python -c "import json; d=json.loads(open('/tmp/b_$dt.json').read().strip().splitlines()[-1]); print('$dt', d['value'])" >> $OUT
done
cat $OUT
