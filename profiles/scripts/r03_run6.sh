set -x
export TMPDIR=/tmp
O=gpurun_out/r3f; mkdir -p $O
( time timeout 600 python -m pytest tests/test_partitioned.py tests/test_gpu_driver.py -m gpu -q -k "partitioned or partition" ) > $O/pytest.log 2>&1; tail -5 $O/pytest.log | cut -c1-400
( time MTB_PART_TIMING=1 timeout 240 python bench.py --partitioned --reads 2000000 --targets 2e9 --steps 3 --warmup 1 --no-parity ) > $O/bench_part.json 2> $O/bench_part.log; grep "partitioned\]" $O/bench_part.log | tail -2; grep -o '"ms_per_step": [0-9.]*' $O/bench_part.json
