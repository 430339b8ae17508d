TAG=${1:-r05_c13}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
MTB_LIB=$R/metabuli_amd/csrc/libmtb_xmerge.so timeout 600 python bench.py --steps 5 --warmup 2 --no-legs --no-cpu --cpu-reads 200000 > $O/${TAG}_bench_merge.json 2> $O/${TAG}_bench_merge.log; grep "stage ms\|parity" $O/${TAG}_bench_merge.log | cut -c1-200
timeout 600 python bench.py --steps 5 --warmup 2 --no-legs --no-cpu --cpu-reads 200000 > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.log; grep "stage ms\|parity" $O/${TAG}_bench_default.log | cut -c1-200
python - <<'PY'
import json,sys
for n in ("merge","default"):
    d=json.load(open(f"gpurun_out/r05_c13/r05_c13_bench_{n}.json"))
    print(n, round(d["ms_per_step"],1), {k:round(v["ms"],1) for k,v in d["kernel_ms"].items() if v["launches"]}, d["config"]["reads_scored_by_generic_kernel"])
PY
