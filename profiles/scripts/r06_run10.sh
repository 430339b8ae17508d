#!/bin/bash
# round 6, GPU call 10: the window form on long reads (MODE 1), in-process A/B at full size (two sub-batches of 100 k reads: 175 queries per tile)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_run10; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "share_a_long or (test_fused_batch and sync_long)" > $O/pytest_subset.txt 2>&1; tail -2 $O/pytest_subset.txt
timeout 900 python bench.py --seq-mode 3 --reads 200000 --read-len 10000 --steps 2 --warmup 1 --no-cpu --cpu-reads 333334 --ab "MTB_JOIN_VARIANT=window;MTB_JOIN_VARIANT=q1w6;MTB_JOIN_VARIANT=window,MTB_JOIN_WIN_QT=256;MTB_JOIN_VARIANT=window,MTB_JOIN_WIN_QT=128;MTB_JOIN_VARIANT=window" > $O/long_bench.json 2> $O/long_bench.log
echo "long rc=$?"; grep -E "A/B |stage ms|parity" $O/long_bench.log | cut -c1-200
cp bench_detail.json $O/long_detail.json 2>/dev/null
