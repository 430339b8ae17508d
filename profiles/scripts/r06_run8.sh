#!/bin/bash
# round 6, GPU call 8: the whole GPU suite on the consolidated state; default bench (held-out leg with the 27 KB tier-2 scorer); 10 M held-out reads.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_run8; mkdir -p $O; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -8 $O/pytest_gpu.txt | cut -c1-200
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu --cpu-reads 200000 > $O/bench.json 2> $O/bench.log
echo "bench rc=$?"; grep -E "stage ms|leg |parity" $O/bench.log | cut -c1-200
cp bench_detail.json $O/bench_detail.json 2>/dev/null
timeout 900 python bench.py --reads-from heldout --steps 3 --warmup 1 --no-legs --no-cpu --cpu-reads 100000 > $O/heldout_bench.json 2> $O/heldout_bench.log
echo "heldout rc=$?"; grep -E "stage ms|parity" $O/heldout_bench.log | cut -c1-200
cp bench_detail.json $O/heldout_detail.json 2>/dev/null
du -sh $O
