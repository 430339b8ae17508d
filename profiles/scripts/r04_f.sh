# round 4, GPU session F (experiment): is the join's extra time on the heavy-tailed workload the overflow list's single counter?
# the same run with tails long enough that (nearly) nothing overflows
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=gpurun_out/r4f; mkdir -p $O; export TMPDIR=/tmp
MTB_TAIL_MIN=112 timeout 400 python bench.py --steps 3 --warmup 1 --no-parity --no-legs > $O/bench_tail112.json 2> $O/bench_tail112.log; echo "rc=$?"; grep "stage ms" $O/bench_tail112.log | cut -c1-300
python - <<'P'
import json
d=json.loads(open('gpurun_out/r4f/bench_tail112.json').read().strip().splitlines()[-1])
print({k:v for k,v in d['kernel_ms'].items() if v['launches']}, d['config']['reads_scored_by_generic_kernel'])
P
