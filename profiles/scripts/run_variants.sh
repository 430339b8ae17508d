# A/B runs of bench.py with environment knobs / flags; prints ms per step and stage times
for v in "--streams 1" "--streams 1"; do
  timeout 400 python bench.py --steps 6 --warmup 3 --cpu-reads 400000 --no-cpu $v > gpurun_out/var.json 2> gpurun_out/var.err
  echo "$v: $(python -c "import json; j=json.load(open('gpurun_out/var.json')); print(round(j['ms_per_step'],1), round(j['value'],1), j['parity_sample']['mismatches'], j['config']['sub_batches_per_step'], {k: round(x['ms'],1) for k,x in j['kernel_ms'].items() if x['ms']>1})") $(grep 'stage ms' gpurun_out/var.err | cut -c1-150)"
done
