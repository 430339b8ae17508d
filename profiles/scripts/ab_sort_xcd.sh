for i in 1 2; do
python bench.py --steps 5 --warmup 3 --no-parity 2>&1 | tail -1 > gpurun_out/ab_xcd_$i.json
MTB_SORT_NO_XCD=1 python bench.py --steps 5 --warmup 3 --no-parity 2>&1 | tail -1 > gpurun_out/ab_noxcd_$i.json
done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/ab_*.json")):
    d=json.loads(open(f).read()); print(f, round(d["ms_per_step"],1), {k:v["ms"] for k,v in d["roofline_all"].items()})
PY
