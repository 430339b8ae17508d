TAG=${1:-r05_c21}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python bench.py --seq-mode 3 --reads 50000 --read-len 10000 --steps 3 --warmup 1 --no-parity --no-legs > $O/${TAG}_long.json 2> $O/${TAG}_long.log; grep "stage ms" $O/${TAG}_long.log | tail -1 | cut -c1-300
MTB_JOIN_VARIANT=q1w6 timeout 600 python bench.py --seq-mode 3 --reads 50000 --read-len 10000 --steps 3 --warmup 1 --no-parity --no-legs > $O/${TAG}_long_q1w6.json 2> $O/${TAG}_long_q1w6.log; grep "stage ms" $O/${TAG}_long_q1w6.log | tail -1 | cut -c1-300
