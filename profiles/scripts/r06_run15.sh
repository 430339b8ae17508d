#!/bin/bash
# round 6, GPU call 15: the wave scan with its fetches running a step ahead (two register sets, vmcnt drained before the first fetch), the scorer's big temporaries
# inside the buffers the join leaves dead (reads of held-out genomes: one sub-batch instead of two)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_run15; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "long_candidate_runs or target_windows or match_and_sort or many_species or deferred_reads_beyond or fused" --timeout 300 > $O/pytest_subset.txt 2>&1; tail -3 $O/pytest_subset.txt
MTB_HOST_TIMING=1 timeout 600 python bench.py --reads-from heldout --steps 3 --warmup 1 --no-legs --no-cpu --cpu-reads 100000 --ab "MTB_SCRATCH_ALIAS=-1" > $O/heldout.json 2> $O/heldout.log
echo "heldout rc=$?"; grep -E "A/B|stage ms|parity|mtb budget: free" $O/heldout.log | cut -c1-260 | tail -12
cp bench_detail.json $O/heldout_detail.json
timeout 500 python bench.py --steps 10 --warmup 3 --no-legs --no-cpu --cpu-reads 100000 > $O/headline.json 2> $O/headline.log
echo "headline rc=$?"; grep -E "stage ms|parity|tuner" $O/headline.log | tail -3 | cut -c1-220
timeout 600 python bench.py --seq-mode 3 --reads 200000 --read-len 10000 --steps 3 --warmup 1 --no-cpu --cpu-reads 333334 > $O/long.json 2> $O/long.log
echo "long rc=$?"; grep -E "stage ms|parity" $O/long.log | cut -c1-200
timeout 600 python bench.py --seq-mode 2 --reads 12500000 --steps 2 --warmup 1 --no-cpu --cpu-reads 100000 > $O/paired.json 2> $O/paired.log
echo "paired rc=$?"; grep -E "stage ms|parity" $O/paired.log | cut -c1-200
