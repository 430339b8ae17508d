# A/B of tuning builds on one box: bash profiles/scripts/ab_libs.sh libmtb.so libmtb_x5.so ...   (extra bench flags in $BENCH_FLAGS)
for rep in 1 2; do
for lib in "$@"; do
MTB_LIB=$PWD/metabuli_amd/csrc/$lib python bench.py --steps 5 --warmup 3 --no-parity $BENCH_FLAGS 2>&1 | tail -1 > gpurun_out/ab_${lib}_$rep.json
done; done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/ab_lib*.json")):
    d=json.loads(open(f).read()); print(f, round(d["ms_per_step"],1), {k:v["ms"] for k,v in d["roofline_all"].items()})
PY
