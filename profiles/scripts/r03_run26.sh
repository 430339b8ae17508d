cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 600 python bench.py --cpu-reads 1000000 --cpu-targets 16e6 --steps 3 --warmup 2 --seq-mode 2 --reads 12500000 > $O/r03_final_bench_paired.json 2> $O/r03_final_bench_paired.log; grep "stage ms" $O/r03_final_bench_paired.log
rm -rf $O/prof_paired && mkdir -p $O/prof_paired
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof_paired -o ks -- python $R/bench.py --steps 3 --warmup 2 --no-parity --seq-mode 2 --reads 12500000 > $R/$O/prof_paired/bench.json 2> $R/$O/prof_paired/bench.log )
python profiles/scripts/rocpd_summary.py $(find $O/prof_paired -name "*.db" | head -1) > $O/r03_final_paired_rocprofv3_kernel_stats.txt 2>&1; head -8 $O/r03_final_paired_rocprofv3_kernel_stats.txt | cut -c1-150
find $O/prof_paired -name "*.db" -size +30M -delete
python - <<'PY'
import json
j=json.load(open("gpurun_out/r03_final_bench_paired.json")); k=j["kernel_ms"]
print(round(j["ms_per_step"],1), round(j["value"],2), {x:round(k[x]["ms"],2) for x in k if k[x]["ms"]>0}, (j.get("parity_full_index") or {}).get("mismatches"), (j.get("parity_sample") or {}).get("mismatches"))
PY
