# Round 5, GPU call 5: kernel trace of the held-out-organism workload (2 M reads) after k_many_sort
TAG=${1:-r05_c5}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
S=/tmp/mtb_prof_scratch; rm -rf $S; mkdir -p $S
( cd /tmp && MTB_MANY_VERBOSE=1 timeout 400 rocprofv3 --kernel-trace --stats -d $S/prof_ks -o ks -- python $R/bench.py --reads-from heldout --reads 2000000 --steps 3 --warmup 1 --no-legs --no-parity > $O/${TAG}_novel_ks.json 2> $O/${TAG}_novel_ks.log )
python profiles/scripts/rocpd_summary.py $(find $S/prof_ks -name "*.db" | head -1) > $O/${TAG}_novel_rocprofv3_kernel_stats.txt 2>&1; grep -v "at::\|rocprim\|rocclr\|k_synth\|k_dir\|k_index" $O/${TAG}_novel_rocprofv3_kernel_stats.txt | head -24 | cut -c1-150
grep "stage ms" $O/${TAG}_novel_ks.log | tail -1 | cut -c1-300
rm -rf $S
