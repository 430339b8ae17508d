"""Synthetic inputs for tests and bench.py: taxonomy dumps,
genomes with genus/species/strain structure, Illumina-/ONT-like reads and the
(value, species)-deduplicated target list of a database.

Pure numpy; no reference code, no GPU.  Everything is seeded.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field

import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
_COMP[:] = ord("N")
for a, b in zip(b"ACGTacgt", b"TGCAtgca"):
    _COMP[a] = b


def revcomp(seq: np.ndarray) -> np.ndarray:
    return _COMP[seq[::-1]]


@dataclass
class Taxonomy:
    """Tiny NCBI-style taxonomy: ids are dense, 1 = root."""
    parent: dict = field(default_factory=dict)
    rank: dict = field(default_factory=dict)
    name: dict = field(default_factory=dict)

    def add(self, tid, parent, rank, name):
        self.parent[tid] = parent
        self.rank[tid] = rank
        self.name[tid] = name

    def write(self, d):
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "nodes.dmp"), "w") as f:
            for t in sorted(self.parent):
                f.write(f"{t}\t|\t{self.parent[t]}\t|\t{self.rank[t]}\t|\t\t|\n")
        with open(os.path.join(d, "names.dmp"), "w") as f:
            for t in sorted(self.parent):
                f.write(f"{t}\t|\t{self.name[t]}\t|\t\t|\tscientific name\t|\n")
        with open(os.path.join(d, "merged.dmp"), "w") as f:
            pass

    def lineage(self, t):
        out = [t]
        while self.parent[t] != t:
            t = self.parent[t]
            out.append(t)
        return out

    def lca(self, a, b):
        la = self.lineage(a)
        sb = set(self.lineage(b))
        for x in la:
            if x in sb:
                return x
        return 1

    def species_of(self, t):
        for x in self.lineage(t):
            if self.rank[x] == "species":
                return x
        return 0


@dataclass
class World:
    tax: Taxonomy
    genomes: list          # list of (strain_taxid, np.uint8 array)
    species: list          # species taxids
    filler_tax_lo: int = 0
    filler_tax_hi: int = 0


def mutate(rng, seq, rate):
    out = seq.copy()
    n = len(seq)
    k = rng.binomial(n, rate)
    if k:
        pos = rng.choice(n, size=k, replace=False)
        # substitute with a different base
        cur = out[pos]
        sub = ACGT[rng.integers(0, 4, size=k)]
        same = sub == cur
        sub[same] = ACGT[(np.searchsorted(ACGT, cur[same]) + 1) % 4]
        out[pos] = sub
    return out


def make_world(seed=1, n_genera=4, species_per_genus=2, strains_per_species=2,
               genome_len=30000, genus_div=0.15, strain_div=0.01, with_euk=True,
               n_filler_species=0, strain_rank="no rank") -> World:
    """root -> {Bacteria, Eukaryota} -> genus -> species -> strain (no rank)."""
    rng = np.random.default_rng(seed)
    tax = Taxonomy()
    tax.add(1, 1, "no rank", "root")
    tax.add(2, 1, "superkingdom", "Bacteria")
    tax.add(3, 1, "superkingdom", "Eukaryota")
    nxt = 4
    genomes, species = [], []
    for g in range(n_genera):
        dom = 3 if (with_euk and g == n_genera - 1) else 2
        gid = nxt; nxt += 1
        tax.add(gid, dom, "genus", f"Genus{g}")
        anc = ACGT[rng.integers(0, 4, size=genome_len)]
        for s in range(species_per_genus):
            sid = nxt; nxt += 1
            tax.add(sid, gid, "species", f"Genus{g} species{s}")
            species.append(sid)
            sp_seq = mutate(rng, anc, genus_div)
            for k in range(strains_per_species):
                tid = nxt; nxt += 1
                tax.add(tid, sid, strain_rank, f"Genus{g} species{s} strain{k}")
                genomes.append((tid, mutate(rng, sp_seq, strain_div)))
    lo = nxt
    for i in range(n_filler_species):
        tax.add(nxt, 2, "species", f"filler{i}")
        nxt += 1
    return World(tax, genomes, species, lo, nxt - 1)


def sample_reads(rng, world: World, n, length=150, err=0.005, frac_random=0.1,
                 with_n=0.0, paired=False, insert=(300, 500), lognormal=False,
                 indel=0.0):
    """Returns (bases, offs[, bases2, offs2], truth) as numpy arrays.
    truth[i] = strain taxid or 0 for random reads."""
    seqs1, seqs2, truth = [], [], []
    for i in range(n):
        if lognormal:
            L = int(np.clip(rng.lognormal(np.log(length), 0.5), 1000, 50000))
        else:
            L = length
        if rng.random() < frac_random:
            r1 = ACGT[rng.integers(0, 4, size=L)]
            r2 = ACGT[rng.integers(0, 4, size=L)]
            truth.append(0)
        else:
            tid, g = world.genomes[rng.integers(0, len(world.genomes))]
            if paired:
                ins = int(rng.integers(insert[0], insert[1]))
                ins = min(ins, len(g))
                st = int(rng.integers(0, len(g) - ins + 1))
                frag = g[st:st + ins]
                if rng.random() < 0.5:
                    frag = revcomp(frag)
                r1 = frag[:L].copy()
                r2 = revcomp(frag)[:L].copy()
            else:
                LL = min(L, len(g))
                st = int(rng.integers(0, len(g) - LL + 1))
                r1 = g[st:st + LL].copy()
                if rng.random() < 0.5:
                    r1 = revcomp(r1)
                r2 = None
            r1 = mutate(rng, r1, err)
            if r2 is not None:
                r2 = mutate(rng, r2, err)
            if indel > 0:
                keep = rng.random(len(r1)) >= indel
                r1 = r1[keep]
            truth.append(tid)
        if with_n > 0 and rng.random() < with_n:
            r1 = r1.copy()
            r1[rng.integers(0, len(r1))] = ord("N")
        seqs1.append(r1)
        if paired:
            seqs2.append(r2)
    def cat(seqs):
        offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([len(s) for s in seqs])
        bases = np.concatenate(seqs).astype(np.uint8) if seqs else np.zeros(0, np.uint8)
        return bases, offs
    b1, o1 = cat(seqs1)
    if paired:
        b2, o2 = cat(seqs2)
        return b1, o1, b2, o2, np.array(truth, dtype=np.int32)
    return b1, o1, np.array(truth, dtype=np.int32)


def dedup_targets(world: World, values_per_genome):
    """IndexCreator's filterKmers<DB_CREATION> (IndexCreator.h:546-580):
    one entry per (value, species); its taxid is the LCA of the contributing
    taxids.  values_per_genome: list of uint64 arrays aligned with
    world.genomes.  Returns (values, taxids) sorted by (value, species, taxid)."""
    vals, tids, sps = [], [], []
    for (tid, _), v in zip(world.genomes, values_per_genome):
        v = np.unique(v)
        vals.append(v)
        tids.append(np.full(len(v), tid, dtype=np.int32))
        sps.append(np.full(len(v), world.tax.species_of(tid), dtype=np.int32))
    vals = np.concatenate(vals); tids = np.concatenate(tids); sps = np.concatenate(sps)
    order = np.lexsort((tids, sps, vals))
    vals, tids, sps = vals[order], tids[order], sps[order]
    # group boundaries on (value, species)
    new = np.ones(len(vals), dtype=bool)
    new[1:] = (vals[1:] != vals[:-1]) | (sps[1:] != sps[:-1])
    starts = np.flatnonzero(new)
    ends = np.append(starts[1:], len(vals))
    out_t = tids[starts].copy()
    multi = np.flatnonzero(ends - starts > 1)
    for gi in multi:
        t = int(tids[starts[gi]])
        for j in range(starts[gi] + 1, ends[gi]):
            t = world.tax.lca(t, int(tids[j]))
        out_t[gi] = t
    ov, osp = vals[starts], sps[starts]
    order = np.lexsort((out_t, osp, ov))
    return ov[order], out_t[order]
