TAG=${1:-r05_c11}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
MTB_LIB=$R/metabuli_amd/csrc/libmtb_xnostore.so timeout 600 python bench.py --steps 3 --warmup 1 --no-legs --no-parity --ab "MTB_JOIN_WIN=0" > $O/${TAG}_bench_nostore.json 2> $O/${TAG}_bench_nostore.log; grep "stage ms\|A/B\|sanity" $O/${TAG}_bench_nostore.log | cut -c1-200
MTB_LIB=$R/metabuli_amd/csrc/libmtb_dbg.so timeout 600 python bench.py --steps 3 --warmup 1 --no-legs --no-parity > $O/${TAG}_bench_dbg.json 2> $O/${TAG}_bench_dbg.log; grep "stage ms\|exits" $O/${TAG}_bench_dbg.log | cut -c1-400
