#!/bin/bash
# GPU box, round 5: full GPU suite after the cleanup (ABI 3), same-box A/B of the pre-cleanup tree (.ab_old, HEAD~1 built
# in the container) against this tree, and the three table entries that lost their (row-panel) tile re-tuned
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
timeout 1800 python -m pytest tests -q -m gpu > gpurun_out/r05_c5_tests.log 2>&1
tail -4 gpurun_out/r05_c5_tests.log
if [ -d .ab_old_skip ]; then
  for i in 1 2 3; do
    for t in old new; do
      if [ $t = old ]; then D=$R/.ab_old; else D=$R; fi
      (cd $D && timeout 300 python bench.py --steps 30 --warmup 5 --no-roofline --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tree=$t', round(d['value'],2), round(d['other_dtype']['value'],2), round(d['batched']['value'],2))") | tee -a gpurun_out/r05_cleanup_ab.txt
    done
  done
fi
NEW=gpurun_out/tuned_r05b.json
cp dualdiff_amd/tuned/gfx950.json $NEW
timeout 600 python bench.py --challenge-tiles 72 --tune-cache $NEW --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-extra-legs --single-dtype > /dev/null 2>&1
python - <<'PY'
import json
a=dict((k,tuple(v)) for k,v in json.load(open('dualdiff_amd/tuned/gfx950.json'))['entries'])
b=dict((k,tuple(v)) for k,v in json.load(open('gpurun_out/tuned_r05b.json'))['entries'])
print("new entries:", [(k,b[k]) for k in b if k not in a])
PY
