set -x
export TMPDIR=/tmp
O=gpurun_out/r3a; mkdir -p $O
( time timeout 420 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "streams or views or borrowed or packed or fused_batch or register_resident" ) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
( time timeout 500 python profiles/scripts/contig_diag.py ) > $O/contig_diag.log 2>&1; cat $O/contig_diag.log | cut -c1-400
( time timeout 400 python bench.py --steps 5 --warmup 2 ) > $O/bench.json 2> $O/bench.log; tail -6 $O/bench.log | cut -c1-600
( time timeout 600 python bench.py --steps 3 --warmup 2 --species 2400 --fixed-total --no-cpu ) > $O/bench_div.json 2> $O/bench_div.log; tail -8 $O/bench_div.log | cut -c1-600
