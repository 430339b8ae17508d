TAG=${1:-r05_c20}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --maxfail=5 -p no:cacheprovider -k "long" > $O/${TAG}_pytest_long.log 2>&1; tail -n 2 $O/${TAG}_pytest_long.log | cut -c1-300
timeout 600 python bench.py --seq-mode 3 --reads 50000 --read-len 10000 --steps 3 --warmup 1 --no-cpu --cpu-reads 166667 --no-legs > $O/${TAG}_long.json 2> $O/${TAG}_long.log; grep "parity\|stage ms" $O/${TAG}_long.log | tail -2 | cut -c1-500
