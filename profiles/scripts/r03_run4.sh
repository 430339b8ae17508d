set -x
export TMPDIR=/tmp
O=gpurun_out/r3d; mkdir -p $O
( time timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "long" ) > $O/pytest.log 2>&1; tail -15 $O/pytest.log
( time timeout 600 python bench.py --no-cpu --steps 2 --warmup 1 --seq-mode 3 --reads 200000 --read-len 10000 ) > $O/bench_long.json 2> $O/bench_long.log; tail -6 $O/bench_long.log | cut -c1-900
