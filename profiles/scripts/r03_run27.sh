cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3y; O=gpurun_out/r3y
for v in merge base; do
  if [ $v = merge ]; then export MTB_LIB=$PWD/metabuli_amd/csrc/libmtb_xmerge.so; else unset MTB_LIB; fi
  timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu --no-parity --species 2400 --fixed-total > $O/div_$v.json 2> $O/div_$v.log
  python - <<PY
import json
j=json.load(open("gpurun_out/r3y/div_$v.json")); k=j["kernel_ms"]
print("$v", round(j["ms_per_step"],1), {x:round(k[x]["ms"],2) for x in ("score","score_fast","segsort","join")}, j["config"].get("reads_scored_by_generic_kernel"))
PY
done
