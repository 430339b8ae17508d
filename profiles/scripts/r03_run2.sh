set -x
export TMPDIR=/tmp
O=gpurun_out/r3b; mkdir -p $O
( time timeout 420 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "streams or views or borrowed or packed" ) > $O/pytest.log 2>&1; tail -6 $O/pytest.log
( time timeout 300 python profiles/scripts/contig_diag.py fence "baseline" "poisoned" ) > $O/contig_diag.log 2>&1; cat $O/contig_diag.log | cut -c1-300
( time timeout 900 python bench.py --steps 3 --warmup 2 --species 2400 --fixed-total --no-cpu ) > $O/bench_div.json 2> $O/bench_div.log; tail -8 $O/bench_div.log | cut -c1-800
