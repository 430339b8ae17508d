#!/bin/bash
# round 6, GPU call 12: k_score_many's staging on pairs (192 vs 320 records), the two-launch k_score_long on long reads
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_run12; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sync_long or sync_xlong or many_species or deferred_reads_beyond" --timeout 300 > $O/pytest_subset.txt 2>&1; tail -2 $O/pytest_subset.txt
timeout 700 python bench.py --seq-mode 2 --reads 12500000 --steps 2 --warmup 1 --no-cpu --cpu-reads 100000 --ab "MTB_MANY_CAP=192;MTB_MANY_CAP=320;MTB_MANY_CAP=192" > $O/paired.json 2> $O/paired.log
echo "paired rc=$?"; grep -E "A/B |stage ms|parity" $O/paired.log | cut -c1-200
timeout 600 python bench.py --seq-mode 3 --reads 200000 --read-len 10000 --steps 3 --warmup 1 --no-cpu --cpu-reads 333334 > $O/long.json 2> $O/long.log
echo "long rc=$?"; grep -E "stage ms|parity" $O/long.log | cut -c1-200
cp bench_detail.json $O/long_detail.json
