cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -f gpurun_out/r03l_order.txt
p() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); o=d.get('other_dtype') or {}; print('$1', d['dtype'], round(d['value'],2), o.get('dtype'), round(o.get('value',0),2))" | tee -a gpurun_out/r03l_order.txt; }
for i in 1 2; do
python bench.py --steps 30 --warmup 5 --no-roofline --no-cpu-baseline --single-dtype --dtype fp16 2>/dev/null | p single
python bench.py --steps 30 --warmup 5 --no-roofline --no-cpu-baseline --single-dtype --dtype bf16 2>/dev/null | p single
python bench.py --steps 30 --warmup 5 --no-roofline --no-cpu-baseline --dtype fp16 2>/dev/null | p both
python bench.py --steps 30 --warmup 5 --no-roofline --no-cpu-baseline --dtype bf16 2>/dev/null | p both
done
