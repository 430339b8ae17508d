TAG=${1:-r05_c18}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
for v in 3 4; do
MTB_LIB=$R/metabuli_amd/csrc/libmtb_xsoexp$v.so timeout 600 python bench.py --seq-mode 3 --reads 50000 --read-len 10000 --steps 2 --warmup 1 --no-parity --no-legs > $O/${TAG}_long_exp$v.json 2> $O/${TAG}_long_exp$v.log; echo "exp $v"; grep "phases\|stage ms" $O/${TAG}_long_exp$v.log | tail -2 | cut -c1-500
done
