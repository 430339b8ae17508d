TAG=${1:-r05_c15}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
MTB_LIB=$R/metabuli_amd/csrc/libmtb_xsoprof.so timeout 600 python bench.py --seq-mode 3 --reads 50000 --read-len 10000 --steps 2 --warmup 1 --no-parity --no-legs > $O/${TAG}_long_soprof.json 2> $O/${TAG}_long_soprof.log; grep "phases\|stage ms" $O/${TAG}_long_soprof.log | tail -2 | cut -c1-500
timeout 600 python bench.py --seq-mode 3 --reads 50000 --read-len 10000 --steps 3 --warmup 1 --no-cpu --cpu-reads 100000 --no-legs > $O/${TAG}_long.json 2> $O/${TAG}_long.log; grep "parity\|stage ms" $O/${TAG}_long.log | tail -2 | cut -c1-500
