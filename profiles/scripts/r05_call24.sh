TAG=${1:-r05_c24}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --maxfail=5 -p no:cacheprovider -k "lockstep or long_candidate_runs or runs_beyond_256 or target_windows" > $O/${TAG}_pytest_subset.log 2>&1; tail -n 2 $O/${TAG}_pytest_subset.log | cut -c1-300
timeout 900 python bench.py --steps 5 --warmup 5 --no-cpu --cpu-reads 200000 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.log; grep "stage ms\|parity\|leg " $O/${TAG}_bench.log | cut -c1-200
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05_c24/r05_c24_bench.json"))
for n,e in d["other_configs"].items(): print(n, round(e["ms_per_step"],1), {k:round(v,1) for k,v in e["stage_ms"].items()})
PY
