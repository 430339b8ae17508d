# the driver's command on the final state of the round
TAG=${1:-r05_last}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
MTB_JOIN_VERBOSE=1 timeout 230 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.log; echo "bench rc=$?"; grep "stage ms\|leg \|parity\|tuned" $O/${TAG}_bench.log | cut -c1-220
