#!/bin/bash
# Run-to-run spread of the join: N processes of the default bench under rocprofv3 with translation counters; per process the
# join's duration next to its UTCL1 hit/miss counts and UTCL2 busy cycles.  Usage: bash profiles/scripts/pmc_join_spread.sh [n]
R=${GRAFT_REPO_ROOT:-$(pwd)}
n=${1:-5}
cd /tmp && export TMPDIR=/tmp
for i in $(seq 1 $n); do
  timeout 600 rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE --kernel-trace --output-format csv \
     -d $R/gpurun_out/spread_$i -- python $R/bench.py --steps 2 --warmup 3 --no-parity > $R/gpurun_out/spread_$i.log 2>&1
  echo "run $i rc=$?"
done
python3 - <<PY
import csv, glob, collections, os
R = "$R"
for d in sorted(glob.glob(R + "/gpurun_out/spread_*")):
    if not os.path.isdir(d): continue
    cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True); kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    if not cc or not kt: print(d, "no output"); continue
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(kt[0])):
        dur[r["Kernel_Name"][:28]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    cnt = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(cc[0])):
        cnt[r["Kernel_Name"][:28]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in dur:
        if not any(s in k for s in ("k_join_dir", "k_radix_scatter", "k_score_fast", "k_extract<2>")): continue
        v = dur[k][-2:]
        line = f"{os.path.basename(d)} {k:30s} ms {sum(v)/len(v):8.3f}"
        for c, vals in sorted(cnt[k].items()):
            vv = vals[-2:]; line += f"  {c.replace('TCP_UTCL1_','').replace('_sum','')} {sum(vv)/len(vv):.4g}"
        print(line)
PY
find $R/gpurun_out -name "*counter_collection.csv" -size +20M -delete
