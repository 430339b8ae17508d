#!/bin/bash
# round 6, GPU call 3: (1) the low-dword window (WIN == 2) at 5..8 waves per SIMD against the 8-byte window and the sector-random join, in one process,
# headline batch + legs; (2) 10 M reads of HELD-OUT genomes as the main workload: kernel stats (rocprofv3) + in-process A/B of the join's forms there.
# Raw profiler output stays in /tmp on the box; only summaries come back.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_run3; mkdir -p $O; export TMPDIR=/tmp
S=/tmp/mtb_prof_scratch; rm -rf $S; mkdir -p $S
AB="MTB_JOIN_VARIANT=window;MTB_JOIN_VARIANT=win32w5;MTB_JOIN_VARIANT=win32w6;MTB_JOIN_VARIANT=win32w7;MTB_JOIN_VARIANT=win32w8;MTB_JOIN_VARIANT=q1w6;MTB_JOIN_VARIANT=window,MTB_JOIN_NO_PREWIN=1;MTB_JOIN_VARIANT=win32w6,MTB_JOIN_COOP_MIN=16;MTB_JOIN_VARIANT=window;MTB_JOIN_VARIANT=win32w6;MTB_JOIN_VARIANT=win32w7"
timeout 1200 python bench.py --steps 5 --warmup 2 --no-cpu --cpu-reads 200000 --ab "$AB" > $O/bench_ab.json 2> $O/bench_ab.log
echo "bench rc=$?"; grep -E "A/B headline|stage ms|leg |parity|join tuned" $O/bench_ab.log | cut -c1-200
cp bench_detail.json $O/bench_ab_detail.json 2>/dev/null
# held-out genomes as the main workload: 10 M reads, kernel stats
AB2="MTB_JOIN_VARIANT=window;MTB_JOIN_VARIANT=win32w6;MTB_JOIN_VARIANT=q1w6;MTB_JOIN_VARIANT=q2w5"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $S/prof_ks -o ks -- python $R/bench.py --reads-from heldout --steps 3 --warmup 1 --no-legs --no-cpu --cpu-reads 100000 --ab "$AB2" > $O/heldout_bench.json 2> $O/heldout_bench.log )
echo "heldout rc=$?"; grep -E "A/B headline|stage ms|parity|join tuned" $O/heldout_bench.log | cut -c1-200
python profiles/scripts/rocpd_summary.py $(find $S/prof_ks -name "*.db" | head -1) > $O/r06_heldout_10M_rocprofv3_kernel_stats.txt 2>&1; head -24 $O/r06_heldout_10M_rocprofv3_kernel_stats.txt | cut -c1-150
rm -rf $S; du -sh $O
