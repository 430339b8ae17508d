#!/usr/bin/env python3
"""profiles/*_pmc_hbm.txt (pmc_summary.py output of the FETCH_SIZE and WRITE_SIZE passes) -> profiles/pmc_traffic.json,
which bench.py reads to fill roofline.traffic for the default workload.
usage: pmc_to_json.py SUMMARY.txt READS READ_LEN TARGETS > profiles/pmc_traffic.json"""
import ast, json, re, sys
names = {"k_score<true, true": "score", "k_join<true>": "join", "k_radix_scatter": "radix_scatter", "k_radix_hist": "radix_hist", "k_extract<2>": "extract_emit"}
out = {}
for line in open(sys.argv[1]):
    if line.startswith("#") or "{" not in line:
        continue
    k, d = line.split("{", 1)
    d = ast.literal_eval("{" + d)
    for pat, nm in names.items():
        if pat in k:
            for c, (avg, n) in d.items():
                out.setdefault(nm, {})[c.lower() + "_kb"] = avg
                out[nm]["launches_profiled"] = n
print(json.dumps({"source": sys.argv[1], "workload": {"reads": int(sys.argv[2]), "read_len": int(sys.argv[3]), "targets": int(sys.argv[4]), "seq_mode": 1},
                  "unit": "KB (1024 B) per launch, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes",
                  "correction": "gfx950: FETCH_SIZE reports half of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section): traffic = 2 x FETCH_SIZE + WRITE_SIZE",
                  "kernels": out}, indent=1))
