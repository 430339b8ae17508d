cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3v; O=gpurun_out/r3v; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "long or fused or score or large" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 600 python bench.py --cpu-reads 20000 --cpu-targets 16e6 --steps 2 --warmup 2 --seq-mode 3 --reads 200000 --read-len 10000 > $O/r03_final_bench_long.json 2> $O/bench_long.log; grep "stage ms" $O/bench_long.log; grep "parity" $O/bench_long.log | cut -c1-140
MTB_LIB=$PWD/metabuli_amd/csrc/libmtb_xlprof.so timeout 400 python bench.py --no-cpu --no-parity --steps 1 --warmup 1 --seq-mode 3 --reads 200000 --read-len 10000 > $O/bench_long_prof.json 2> $O/bench_long_prof.log; grep "k_score_long phases" $O/bench_long_prof.log | tail -1
rm -rf $O/prof_long && mkdir -p $O/prof_long
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof_long -o ks -- python $R/bench.py --steps 3 --warmup 2 --no-parity --seq-mode 3 --reads 200000 --read-len 10000 > $R/$O/prof_long/bench.json 2> $R/$O/prof_long/bench.log )
python profiles/scripts/rocpd_summary.py $(find $O/prof_long -name "*.db" | head -1) > $O/r03_final_long_rocprofv3_kernel_stats.txt 2>&1; head -9 $O/r03_final_long_rocprofv3_kernel_stats.txt | cut -c1-150
find $O/prof_long -name "*.db" -size +30M -delete
timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu --species 2400 --fixed-total > $O/r03_final_bench_diversity_2400species.json 2> $O/bench_div.log; grep "stage ms" $O/bench_div.log; grep "parity" $O/bench_div.log | cut -c1-140
python - <<'PY'
import json
for f in ("r03_final_bench_long","r03_final_bench_diversity_2400species"):
    j=json.load(open(f"gpurun_out/r3v/{f}.json")); k=j["kernel_ms"]
    print(f, round(j["ms_per_step"],1), round(j["value"],2), {x:round(k[x]["ms"],2) for x in k if k[x]["ms"]>0}, (j.get("parity_full_index") or {}).get("mismatches"), (j.get("parity_sample") or {}).get("mismatches"), j["config"].get("reads_scored_by_generic_kernel"))
PY
