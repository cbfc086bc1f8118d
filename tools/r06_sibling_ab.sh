#!/bin/bash
# GPU box: the drop-in legs with and without the sibling overlap of the public forward()s (DD_SIBLING_OVERLAP), alternating.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; OUT=gpurun_out/r06_sibling_ab.txt; rm -f $OUT
for i in 1 2 3; do
  for v in 1 0; do
    DD_SIBLING_OVERLAP=$v python bench.py --steps 30 --warmup 5 --single-dtype --no-roofline --no-cpu-baseline --batched-scenes 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
g=lambda k: d.get(k) or {}
print('DD_SIBLING_OVERLAP=$v fused %.2f  dropin %.2f (%.3f of fused)  dropin_varlen %.2f (%.3f of dropin, captures %s)  unipc20 %.2f  finite %s %s' % (d['value'], g('dropin').get('value',0), g('dropin').get('vs_fused',0), g('dropin_varlen').get('value',0), g('dropin_varlen').get('vs_dropin',0), g('dropin_varlen').get('captures'), g('unipc20').get('value',0), g('dropin').get('outputs_finite'), g('dropin_varlen').get('outputs_finite')))" | tee -a $OUT
  done
done
