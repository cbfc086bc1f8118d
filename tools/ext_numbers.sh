cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for a in "--scenes 4" "--scenes 2" "--frames 8" "--frames 8 --fp8-weights --lora-rank 4" "--fp8-weights weights"; do
  python bench.py --steps 20 --warmup 3 --single-dtype --no-roofline --no-cpu-baseline $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$a', round(d['value'],2), round(d['ms_per_step'],3), d['outputs_finite'])" | tee -a gpurun_out/r03k_ext.txt
done
