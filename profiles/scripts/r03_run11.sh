set -x
export TMPDIR=/tmp
O=gpurun_out/r3k; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x ) > $O/pytest.log 2>&1; tail -5 $O/pytest.log | cut -c1-300
( time timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu --cpu-targets 16e6 ) > $O/bench.json 2> $O/bench.log; grep -E "stage ms" $O/bench.log
( time timeout 400 python bench.py --no-cpu --cpu-targets 16e6 --steps 3 --warmup 1 --seq-mode 2 --reads 12500000 ) > $O/bench_paired.json 2> $O/bench_paired.log; grep -E "stage ms" $O/bench_paired.log
( time timeout 900 python bench.py --steps 3 --warmup 2 --species 2400 --fixed-total --no-cpu ) > $O/bench_div.json 2> $O/bench_div.log; grep -E "stage ms" $O/bench_div.log
python - <<'PY'
import json
for f in ("bench","bench_paired","bench_div"):
    j=json.loads(open(f'gpurun_out/r3k/{f}.json').read().strip().splitlines()[-1])
    print(f, round(j['ms_per_step'],1), j['config']['reads_scored_by_generic_kernel'], {k:(round(v['ms'],2),v['launches']) for k,v in j['kernel_ms'].items() if v['launches']}, (j.get('parity_sample') or {}).get('mismatches'), (j.get('parity_full_index') or {}).get('mismatches'))
PY
