#!/bin/bash
# GPU box: the round's closing check — smoke(), the whole GPU suite, the driver's default bench command.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -2
rm -f gpurun_out/r06_parity.csv
timeout 1500 python -m pytest tests -q -m gpu --durations=8 > gpurun_out/r06_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r06_tests.log
tail -14 gpurun_out/r06_tests.log
python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err
wc -c gpurun_out/r06_bench_default.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench_default.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','ms_per_step','steps','warmup','dtype','env')})
for k in ('other_dtype','batched','unipc20','dropin','dropin_varlen'):
    v=d.get(k) or {}
    print(k, {kk:v.get(kk) for kk in ('value','vs_fused','vs_dropin','captures','dtype') if kk in v})
print('roofline', {kk:d['roofline'].get(kk) for kk in ('kernel','frac','avg_us','traffic')})
PY
