#!/bin/bash
# round 6, GPU call 5: k_join_win WITHOUT windows (rnd8: the sector-random form at 8 waves per SIMD) against q1w6 on the sparse legs (pairs, held-out
# reads), on long reads (configs[2], MODE 1) and on 12.5 M pairs (configs[3] shape); the window forms on the pairs.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_run5; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "windows_staged or share_a_long" > $O/pytest_subset.txt 2>&1; tail -2 $O/pytest_subset.txt
AB="MTB_JOIN_VARIANT=rnd8;MTB_JOIN_VARIANT=q1w6;MTB_JOIN_VARIANT=win;MTB_JOIN_VARIANT=winw7;MTB_JOIN_VARIANT=win32w8;MTB_JOIN_VARIANT=rnd8"
timeout 1200 python bench.py --steps 3 --warmup 2 --no-parity --ab "$AB" > $O/bench_ab.json 2> $O/bench_ab.log
echo "bench rc=$?"; grep -E "A/B |stage ms|leg " $O/bench_ab.log | cut -c1-200
cp bench_detail.json $O/bench_ab_detail.json 2>/dev/null
timeout 900 python bench.py --seq-mode 3 --reads 200000 --read-len 10000 --steps 2 --warmup 1 --no-parity --ab "MTB_JOIN_VARIANT=rnd8;MTB_JOIN_VARIANT=q1w6;MTB_JOIN_VARIANT=rnd8" > $O/long_bench.json 2> $O/long_bench.log
echo "long rc=$?"; grep -E "A/B |stage ms" $O/long_bench.log | cut -c1-200
cp bench_detail.json $O/long_detail.json 2>/dev/null
timeout 900 python bench.py --seq-mode 2 --reads 12500000 --steps 2 --warmup 1 --no-parity --ab "MTB_JOIN_VARIANT=win;MTB_JOIN_VARIANT=winw7;MTB_JOIN_VARIANT=q1w6;MTB_JOIN_VARIANT=rnd8" > $O/paired_bench.json 2> $O/paired_bench.log
echo "paired rc=$?"; grep -E "A/B |stage ms" $O/paired_bench.log | cut -c1-200
cp bench_detail.json $O/paired_detail.json 2>/dev/null
du -sh $O
