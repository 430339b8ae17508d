set -x
export TMPDIR=/tmp
O=gpurun_out/r3j; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_partitioned.py -m gpu -q -x ) > $O/pytest.log 2>&1; tail -5 $O/pytest.log | cut -c1-300
( time timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu --cpu-targets 16e6 ) > $O/bench_msd.json 2> $O/bench_msd.log; grep -E "stage ms" $O/bench_msd.log
( time MTB_SORT_LSD=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu --cpu-targets 16e6 ) > $O/bench_lsd.json 2> $O/bench_lsd.log; grep -E "stage ms" $O/bench_lsd.log
python - <<'PY'
import json
for f in ("msd","lsd"):
    j=json.loads(open(f'gpurun_out/r3j/bench_{f}.json').read().strip().splitlines()[-1])
    print(f, j['ms_per_step'], {k:(round(v['ms'],2),v['launches']) for k,v in j['kernel_ms'].items() if v['launches']})
PY
