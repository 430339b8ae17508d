#!/bin/bash
# round 6, GPU call 4: k_join_win (kernels_join_win.h: low-dword window, 32-bit offsets, tiles without a window through the sector-random form)
# against the in-kernel window forms and the sector-random join, one process; then the same on 10 M reads of held-out genomes.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_run4; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "windows_staged or share_a_long" > $O/pytest_subset.txt 2>&1; tail -2 $O/pytest_subset.txt
AB="MTB_JOIN_VARIANT=win;MTB_JOIN_VARIANT=winw7;MTB_JOIN_VARIANT=winw6;MTB_JOIN_VARIANT=win32w8;MTB_JOIN_VARIANT=q1w6;MTB_JOIN_VARIANT=win,MTB_JOIN_COOP_MIN=16;MTB_JOIN_VARIANT=win,MTB_JOIN_COOP_MIN=64;MTB_JOIN_VARIANT=win;MTB_JOIN_VARIANT=win32w8"
timeout 1200 python bench.py --steps 5 --warmup 2 --no-cpu --cpu-reads 200000 --ab "$AB" > $O/bench_ab.json 2> $O/bench_ab.log
echo "bench rc=$?"; grep -E "A/B headline|stage ms|leg |parity|join tuned" $O/bench_ab.log | cut -c1-200
cp bench_detail.json $O/bench_ab_detail.json 2>/dev/null
AB2="MTB_JOIN_VARIANT=win;MTB_JOIN_VARIANT=winw7;MTB_JOIN_VARIANT=q1w6"
timeout 900 python bench.py --reads-from heldout --steps 3 --warmup 1 --no-legs --no-cpu --cpu-reads 100000 --ab "$AB2" > $O/heldout_bench.json 2> $O/heldout_bench.log
echo "heldout rc=$?"; grep -E "A/B headline|stage ms|parity|join tuned" $O/heldout_bench.log | cut -c1-200
cp bench_detail.json $O/heldout_detail.json 2>/dev/null
du -sh $O
