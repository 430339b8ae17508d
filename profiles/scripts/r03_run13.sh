cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3m; O=gpurun_out/r3m
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu --cpu-targets 16e6 > $O/bench.json 2> $O/bench.log; grep "stage ms" $O/bench.log
timeout 400 python bench.py --cpu-reads 1000000 --cpu-targets 16e6 --steps 3 --warmup 2 --seq-mode 2 --reads 12500000 > $O/bench_paired.json 2> $O/bench_paired.log; grep "stage ms\|parity" $O/bench_paired.log | cut -c1-250
python - <<'PY'
import json
for f in ("bench","bench_paired"):
    try:
        j=json.load(open(f"gpurun_out/r3m/{f}.json")); k=j["kernel_ms"]
        print(f, round(j["ms_per_step"],1), {x:round(k[x]["ms"],2) for x in ("score","score_fast","join","radix_scatter","extract_emit")}, j.get("parity_full_index",{}).get("mismatches"))
    except Exception as e: print(f, "ERR", e)
PY
