set -x
export TMPDIR=/tmp
export MTB_LSLOT_VERBOSE=1
O=gpurun_out/r3e; mkdir -p $O
( time timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "long" ) > $O/pytest.log 2>&1; tail -25 $O/pytest.log | cut -c1-300
( time timeout 600 python bench.py --no-cpu --steps 2 --warmup 1 --seq-mode 3 --reads 200000 --read-len 10000 ) > $O/bench_long.json 2> $O/bench_long.log; grep -v "parity" $O/bench_long.log | tail -12 | cut -c1-400; grep parity $O/bench_long.log | cut -c1-200
