# rocprofv3 kernel stats of the paired-end and long-read configurations (BASELINE configs[3] per-GPU share, configs[2]).
# usage (GPU box): bash profiles/scripts/kernel_stats_other_configs.sh TAG
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
for cfg in "paired --seq-mode 2 --reads 12500000" "long --seq-mode 3 --reads 200000 --read-len 10000"; do
  set -- $cfg; name=$1; shift
  rm -rf $O/prof_$name && mkdir -p $O/prof_$name
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d $R/$O/prof_$name -o ks -- python $R/bench.py --steps 3 --warmup 2 --no-parity "$@" > $R/$O/prof_$name/bench.json 2> $R/$O/prof_$name/bench.log )
  python profiles/scripts/rocpd_summary.py $(find $O/prof_$name -name "*.db" | head -1) > $O/${TAG}_${name}_rocprofv3_kernel_stats.txt 2>&1
  head -16 $O/${TAG}_${name}_rocprofv3_kernel_stats.txt
  find $O/prof_$name -name "*.db" -size +30M -delete
done
