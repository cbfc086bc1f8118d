#!/bin/bash
# GPU box: re-tune every 3x3-conv entry of the tracked tile table (the other entries stay), then A/B old vs new table.
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out; export TMPDIR=/tmp
python3 - <<'PY'
import json
t = json.load(open("dualdiff_amd/tuned/gfx950.json"))
k = [k for k in t if isinstance(t[k], list)][0]
before = len(t[k])
t[k] = [e for e in t[k] if not e[0].startswith("('c'")]
print("entries", before, "->", len(t[k]))
json.dump(t, open("gpurun_out/tuned_noconv.json", "w"))
PY
rm -f gpurun_out/tuned_conv_new.json
DD_TUNE_TABLE=$PWD/gpurun_out/tuned_noconv.json timeout 2400 python bench.py --tune-cache gpurun_out/tuned_conv_new.json --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-extra-legs 2>gpurun_out/retune_conv.err | tail -1 | cut -c1-200
ls -la gpurun_out/tuned_conv_new.json
OUT=gpurun_out/r05_retune_conv_ab.txt; rm -f $OUT
for i in 1 2 3; do
  for t in old new; do
    if [ $t = new ]; then export DD_TUNE_TABLE=$PWD/gpurun_out/tuned_conv_new.json; else unset DD_TUNE_TABLE; fi
    timeout 300 python bench.py --steps 30 --warmup 5 --no-roofline --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('table=$t', round(d['value'],2), round(d['other_dtype']['value'],2), round(d['batched']['value'],2), d['outputs_finite'])" | tee -a $OUT
  done
done
