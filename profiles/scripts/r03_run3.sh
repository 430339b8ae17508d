set -x
export TMPDIR=/tmp
O=gpurun_out/r3c; mkdir -p $O
( time timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "streams or views or borrowed or packed" ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
( time timeout 300 python profiles/scripts/contig_diag.py "fences only" "system-scope" "contig + clear by kernel" ) > $O/contig_diag.log 2>&1; cat $O/contig_diag.log | cut -c1-250
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/alloc_probe profiles/scripts/alloc_probe.hip && ( time timeout 300 /tmp/alloc_probe ) > $O/alloc_probe.log 2>&1; cat $O/alloc_probe.log
( time timeout 900 python bench.py --steps 3 --warmup 2 --species 2400 --fixed-total --no-cpu ) > $O/bench_div.json 2> $O/bench_div.log; tail -5 $O/bench_div.log | cut -c1-1200
