set -x
export TMPDIR=/tmp
O=gpurun_out/r3h; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -q ) > $O/pytest_all.log 2>&1; tail -12 $O/pytest_all.log | cut -c1-300
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) 2>&1 | tail -3
( time timeout 400 python bench.py --steps 5 --warmup 2 ) > $O/bench.json 2> $O/bench.log; grep "stage ms" $O/bench.log
( time timeout 400 python bench.py --no-cpu --steps 3 --warmup 1 --seq-mode 2 --reads 12500000 ) > $O/bench_paired.json 2> $O/bench_paired.log; grep -E "stage ms|parity" $O/bench_paired.log | cut -c1-300
