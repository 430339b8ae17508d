cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3n; O=gpurun_out/r3n
MTB_HOST_TIMING=1 timeout 400 python profiles/scripts/e2e_driver.py 30e6 200e6 64 2000000,10000000 > $O/e2e.txt 2>&1; grep -v "^mtb_classify_batch_packed" $O/e2e.txt | tail -12; grep "^mtb_classify_batch_packed" $O/e2e.txt | tail -22 | head -8; grep "^mtb_classify_batch_packed" $O/e2e.txt | tail -3
MTB_LIB=$PWD/metabuli_amd/csrc/libmtb_xmerge.so timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu --cpu-targets 16e6 --no-parity > $O/bench_merge.json 2> $O/bench_merge.log; grep "stage ms" $O/bench_merge.log
timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu --cpu-targets 16e6 --no-parity > $O/bench_base.json 2> $O/bench_base.log; grep "stage ms" $O/bench_base.log
python - <<'PY'
import json
for f in ("bench_merge","bench_base"):
    try:
        j=json.load(open(f"gpurun_out/r3n/{f}.json")); k=j["kernel_ms"]
        print(f, round(j["ms_per_step"],1), {x:round(k[x]["ms"],2) for x in ("score","score_fast","join","radix_scatter","extract_emit")})
    except Exception as e: print(f, "ERR", e)
PY
