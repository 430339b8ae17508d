#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/.  Run in the build
container (needs /root/reference for the reference-derived tables):

    python tests/golden/make_golden.py

1. ref_codon_tables.txt   -- output of oracle/_ref/ref_codon_dump, i.e. the
   reference's own GeneticCode.h compiled verbatim (nuc2aa, nuc2num, base maps).
2. ref_hamming_tables.json -- the numeric literals of hammingLookup and
   HAMMING_LUT0..7 parsed out of src/commons/KmerMatcher.h:66-158 (data, not code).
3. toy_*.npz              -- small end-to-end vectors produced by the ORACLE
   (oracle/liboracle.so) on seeded synthetic inputs: reads, database arrays,
   sorted query k-mers, sorted matches, per-read results.  They pin the oracle
   against regressions and give the GPU tests a fixed target; they do NOT pin
   the oracle against the reference (no reference build exists).
"""
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference/src/commons"


def ref_tables():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "all"], stdout=subprocess.DEVNULL)
    shutil.copy(os.path.join(ROOT, "oracle", "_ref", "ref_codon_tables.txt"), os.path.join(HERE, "ref_codon_tables.txt"))
    src = open(os.path.join(REF, "KmerMatcher.h")).read()
    out = {}
    m = re.search(r"hammingLookup\[8\]\[8\]\s*=\s*\{(.*?)\};", src, re.S)
    out["hammingLookup"] = [int(x) for x in re.findall(r"\d+", m.group(1))]
    for k in range(8):
        m = re.search(r"HAMMING_LUT%d\[64\]\s*=\s*\{(.*?)\};" % k, src, re.S)
        body = re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S)
        out["HAMMING_LUT%d" % k] = [int(x) for x in re.findall(r"\d+", body)]
        assert len(out["HAMMING_LUT%d" % k]) == 64
    assert len(out["hammingLookup"]) == 64
    json.dump(out, open(os.path.join(HERE, "ref_hamming_tables.json"), "w"))


def toy_vectors():
    from helpers import Oracle, build_toy_db, default_params
    from metabuli_amd import synth
    orc = Oracle()
    sets = {"toy_sync_se": dict(syncmer=1, paired=False), "toy_dense_pe": dict(syncmer=0, paired=True),
            "toy_oldfmt_pe": dict(syncmer=0, paired=True, kmer_format=1), "toy_sync_long": dict(syncmer=1, paired=False, seq_mode=3)}
    for name, kw in sets.items():
        if os.path.exists(os.path.join(HERE, name + ".npz")) and "--all" not in sys.argv:
            continue                              # committed vectors are only rewritten on request
        seq_mode = kw.get("seq_mode", 2 if kw["paired"] else 1)
        p = default_params(seq_mode=seq_mode, syncmer=kw["syncmer"], kmer_format=kw.get("kmer_format", 2))
        world = synth.make_world(seed=21, n_genera=3, species_per_genus=2, strains_per_species=2, genome_len=6000)
        d = tempfile.mkdtemp()
        vals, tids = build_toy_db(orc, world, p, d)
        tax = orc.load_taxonomy(os.path.join(d, "taxonomy"))
        db = orc.open_db(d, tax, p)
        if seq_mode == 3:
            out = synth.sample_reads(np.random.default_rng(5), world, 12, length=2500, err=0.04, with_n=0.15, lognormal=True)
        else:
            out = synth.sample_reads(np.random.default_rng(5), world, 60, length=150, err=0.01, with_n=0.15, paired=kw["paired"])
        if kw["paired"]:
            b1, o1, b2, o2, truth = out
        else:
            b1, o1, truth = out
            b2 = np.zeros(0, np.uint8); o2 = np.zeros(0, np.uint64)
        R = orc.classify(db, tax, p, b1, o1, b2 if kw["paired"] else None, o2 if kw["paired"] else None)
        nodes = np.array([[t, world.tax.parent[t]] for t in sorted(world.tax.parent)], dtype=np.int32)
        ranks = np.array([world.tax.rank[t] for t in sorted(world.tax.parent)])
        names = np.array([world.tax.name[t] for t in sorted(world.tax.parent)])
        np.savez_compressed(os.path.join(HERE, name + ".npz"), syncmer=kw["syncmer"], paired=int(kw["paired"]), seq_mode=seq_mode,
                            kmer_format=kw.get("kmer_format", 2), bases=b1, offs=o1,
                            bases2=b2, offs2=o2, db_values=vals, db_taxids=tids, tax_nodes=nodes, tax_ranks=ranks, tax_names=names,
                            kmers=R["kmers"], matches=R["matches"], results=R["results"], tc_tax=R["tc_tax"], tc_cnt=R["tc_cnt"],
                            qlen=R["qlen"], qlen2=R["qlen2"], diffidx=orc.diffidx_encode(vals))


if __name__ == "__main__":
    # the reference leaves the table entries of base codes 4..6 uninitialised: their dump is garbage that changes from
    # run to run (the tests read the entries of codes 0..3 and 7 only), so the committed table files are only rewritten
    # with --all
    if "--all" in sys.argv or not os.path.exists(os.path.join(HERE, "ref_codon_tables.txt")):
        ref_tables()
    toy_vectors()
    print("golden fixtures written to", HERE)
