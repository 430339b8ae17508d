TAG=${1:-r05_c25}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 400 python bench.py --steps 5 --warmup 5 --no-legs --no-cpu --cpu-reads 200000 > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.log; grep "stage ms\|parity" $O/${TAG}_bench_default.log | cut -c1-200
timeout 300 python bench.py --reads-from heldout --reads 2000000 --steps 3 --warmup 4 --no-legs --no-parity > $O/${TAG}_novel_default.json 2> $O/${TAG}_novel_default.log; grep "stage ms" $O/${TAG}_novel_default.log | cut -c1-200
MTB_LIB=$R/metabuli_amd/csrc/libmtb_xls32.so timeout 300 python bench.py --reads-from heldout --reads 2000000 --steps 3 --warmup 4 --no-legs --no-parity > $O/${TAG}_novel_ls32.json 2> $O/${TAG}_novel_ls32.log; grep "stage ms" $O/${TAG}_novel_ls32.log | cut -c1-200
