#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of profiles/scripts/pmc_sq.sh (passes d and e): per kernel
the mean KB per launch over the profiled bench run.  Usage: make_pmc_traffic.py gpurun_out/pmc_<tag> <reads> <read_len> <targets> <seq_mode> <label> [workload key: profiles/pmc_traffic_<key>.json, as bench.py names its workloads]"""
import csv, glob, json, re, sys
base, reads, read_len, targets, seq_mode, label = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
wl_key = sys.argv[7] if len(sys.argv) > 7 else "default"
NAMES = [("k_seg_order", "segsort"), ("k_score_long<1024", "score_fast"), ("k_score_many", "score_many"), ("k_join_dir", "join"), ("k_join<", "join"), ("k_score_fast", "score_fast"), ("k_score<", "score"), ("k_radix_scatter", "radix_scatter"),
         ("k_radix_hist", "radix_hist"), ("k_extract<2>", "extract_emit"), ("k_extract<1>", "extract_emit"), ("k_extract<0>", "extract_count")]
out = {}
for suffix, counter, key in (("_d", "FETCH_SIZE", "fetch_size_kb"), ("_e", "WRITE_SIZE", "write_size_kb")):
    acc = {}
    for f in glob.glob(base + suffix + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            for pat, name in NAMES:
                if pat in r["Kernel_Name"]:
                    acc.setdefault(name, []).append(float(r["Counter_Value"]))
                    break
    for name, vals in acc.items():
        # the bench launches setup kernels of the same name (target extraction): keep the launches of the big batch = the largest ones
        vals = sorted(vals, reverse=True)
        top = [v for v in vals if v > 0.5 * vals[0]]
        out.setdefault(name, {})[key] = sum(top) / len(top)
        out[name]["launches_profiled"] = len(top)
json.dump({"source": label, "workload": {"reads": reads, "read_len": read_len, "targets": targets, "seq_mode": seq_mode, "key": wl_key},
           "unit": "KB (1024 B) per launch, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes",
           "correction": "gfx950: FETCH_SIZE reports half of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section): traffic = 2 x FETCH_SIZE + WRITE_SIZE "
                         "(calibrated for coalesced streams only; for the sector-random reads of the join it is an upper bound)",
           "kernels": out}, open("profiles/pmc_traffic.json" if wl_key == "default" else f"profiles/pmc_traffic_{wl_key}.json", "w"), indent=1)
print(json.dumps(out, indent=1))
