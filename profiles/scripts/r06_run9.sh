#!/bin/bash
# round 6, GPU call 9: the GPU suite (per-test time limit), default bench, 10 M held-out reads, long reads at full size (two sub-batches now?), store-side builds.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_run9; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --timeout 400 > $O/pytest_gpu.txt 2>&1; tail -6 $O/pytest_gpu.txt | cut -c1-200
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu --cpu-reads 200000 > $O/bench.json 2> $O/bench.log
echo "bench rc=$?"; grep -E "stage ms|leg |parity" $O/bench.log | cut -c1-200
cp bench_detail.json $O/bench_detail.json 2>/dev/null
timeout 600 python bench.py --reads-from heldout --steps 3 --warmup 1 --no-legs --no-cpu --cpu-reads 100000 > $O/heldout_bench.json 2> $O/heldout_bench.log
echo "heldout rc=$?"; grep -E "stage ms|parity" $O/heldout_bench.log | cut -c1-200
cp bench_detail.json $O/heldout_detail.json 2>/dev/null
timeout 600 python bench.py --seq-mode 3 --reads 200000 --read-len 10000 --steps 2 --warmup 1 --no-cpu --cpu-reads 333334 > $O/long_bench.json 2> $O/long_bench.log
echo "long rc=$?"; grep -E "stage ms|parity" $O/long_bench.log | cut -c1-200
cp bench_detail.json $O/long_detail.json 2>/dev/null
for lib in libmtb.so libmtb_xnostore.so libmtb_xhalfstore.so libmtb_xplainstore.so; do
  MTB_JOIN_VARIANT=window MTB_LIB=$R/metabuli_amd/csrc/$lib timeout 300 python bench.py --steps 4 --warmup 2 --no-parity --no-legs > $O/store_$lib.json 2> $O/store_$lib.log
  echo "== $lib rc=$?"; grep -E "stage ms" $O/store_$lib.log | cut -c1-160
done
du -sh $O
