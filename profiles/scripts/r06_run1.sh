#!/bin/bash
# round 6, GPU call 1: the window join with its tile windows bounded BEFORE the launch (k_join_tile_win) against round 5's in-kernel window and the
# sector-random join, in ONE process (bench.py --ab); phase cycles of both window forms (profiling build).
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r06_run1; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "windows_staged or share_a_long or (test_fused_batch and sync_se)" > $O/pytest_subset.txt 2>&1; tail -3 $O/pytest_subset.txt
timeout 900 python bench.py --steps 5 --warmup 2 --no-legs --no-cpu --cpu-reads 200000 --ab "MTB_JOIN_NO_PREWIN=1;MTB_JOIN_VARIANT=q1w6" > $O/bench_ab.json 2> $O/bench_ab.log
grep -E "A/B|stage ms|join tuned|parity|headline line" $O/bench_ab.log
cp bench_detail.json $O/bench_ab_detail.json 2>/dev/null
for v in "MTB_JOIN_VARIANT=window" "MTB_JOIN_VARIANT=window MTB_JOIN_NO_PREWIN=1"; do
  tag=$(echo "$v" | tr ' =' '__')
  env $v MTB_LIB=$R/metabuli_amd/csrc/libmtb_prof.so timeout 600 python bench.py --steps 3 --warmup 1 --no-legs --no-parity > $O/prof_$tag.json 2> $O/prof_$tag.log
  echo "== $v"; grep -E "phase cycles|stage ms" $O/prof_$tag.log
done
