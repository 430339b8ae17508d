set -x
export TMPDIR=/tmp
O=gpurun_out/r3i; mkdir -p $O
( time timeout 500 python bench.py --steps 3 --warmup 1 --cpu-targets 1e9 ) > $O/bench_cpu1g.json 2> $O/bench_cpu1g.log; tail -4 $O/bench_cpu1g.log | cut -c1-300
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r3i/bench_cpu1g.json').read().strip().splitlines()[-1])
print(json.dumps(j['cpu_baseline'])[:1500])
PY
