#!/usr/bin/env python3
"""bench.py -- `metabuli classify` hot path on MI355X: Mreads/s (+ Gbp/s) with
the reads and the target index already resident in HBM.

One "step" = one pass of the whole hot path (extract -> radix sort -> join
against the resident index -> per-read scoring) over one batch of synthetic
reads (BASELINE.json configs[1]: 10 M x 150 bp single-end vs a GTDB-scale
synthetic index).  One process per GPU; reads are sharded, the index is
replicated, there is no data-path collective (weak scaling).

The workload (round 4): reads drawn from 2400 genomes (0.6 x coverage -- not the
62 x of 24 genomes, which is this engine's best case and is reported beside it as
`best_case`), an index whose candidate runs are heavy-tailed (conserved segments
shared by the genomes at Zipf-like prevalence, multiplied by further species:
runs up to ~10^4 where the reads hit them; `run_lengths` in the line), and --
after the timed region, outside it -- short legs of the other two single-GPU
configurations (configs[3]'s per-GPU shape: read pairs; configs[2]: long reads)
with parity samples of their own (`other_configs`).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8 ...

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_GBS = 8000.0        # HBM3E peak of one MI355X (MI355X_MICROARCH.md)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


# --------------------------------------------------------------------------------------------------------------------
# synthetic world
# --------------------------------------------------------------------------------------------------------------------
def build_world(seed, n_species, genome_len, n_filler_species):
    from metabuli_amd import synth
    return synth.make_world(seed=seed, n_genera=max(1, n_species // 4), species_per_genus=4, strains_per_species=1,
                            genome_len=genome_len, n_filler_species=n_filler_species)


# conserved segments: (prevalence = fraction of the genomes that carry a segment of the class, segments of the class).  With 2400 genomes
# a segment of prevalence 0.55 gives candidate runs of ~1000 species for the amino-acid 8-mers of its coding frame (78 % of the copies
# keep an 8-mer), prevalence 1 gives ~1900; about 1 % of a batch's query metamers meet such a run.
CONSERVED_CLASSES = ((1.0, 10), (0.55, 150), (0.25, 20), (0.1, 20))
_GENETIC_CODE_TCAG = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"      # standard code, codons in TCAG order


def _codon_tables(torch, dev):
    """codon index = 16 b0 + 4 b1 + b2 with bases in A C G T order (0..3).  -> (syn [64, 6] synonymous codons of every sense codon,
    n_syn [64], sense [61] the sense codons)"""
    tcag = "TCAG"
    aa_of = [None] * 64
    for i, aa in enumerate(_GENETIC_CODE_TCAG):
        b = (tcag[i // 16], tcag[(i // 4) % 4], tcag[i % 4])
        aa_of[16 * "ACGT".index(b[0]) + 4 * "ACGT".index(b[1]) + "ACGT".index(b[2])] = aa
    syn = torch.zeros((64, 6), dtype=torch.int64, device=dev); n_syn = torch.ones(64, dtype=torch.int64, device=dev)
    for c in range(64):
        same = [d for d in range(64) if aa_of[d] == aa_of[c]] if aa_of[c] != "*" else [c]
        n_syn[c] = len(same)
        syn[c, : len(same)] = torch.tensor(same, device=dev)
    sense = torch.tensor([c for c in range(64) if aa_of[c] != "*"], device=dev)
    return syn, n_syn, sense


def build_world_fast(torch, dev, seed, n_species, genome_len, n_filler_species, conserved=True, p_syn=0.7, p_nonsyn=0.03, seg_len=999, n_heldout=0, heldout_div=0.075):
    """Same shape as synth.make_world (root -> {Bacteria, Eukaryota} -> genus -> 4 species -> 1 strain, genus divergence 15 %,
    strain divergence 1 %) for THOUSANDS of genomes: sequences are drawn and mutated on the device (Bernoulli substitutions).
    conserved: every genome additionally carries protein-coding segments of a common pool (CONSERVED_CLASSES: 333 codons each, present in
    a class-dependent fraction of the genomes) at random places, every copy with 70 % of its codons redrawn among the synonymous ones
    and 3 % replaced by another amino acid's -- core genes as real databases hold them: the amino-acid 8-mers of the coding frame are
    shared by hundreds to thousands of species (candidate runs of that length in the index, where reads of ANY genome hit them) while
    their DNA differs from species to species, so that a long run costs its scan, not thousands of matches.
    n_heldout: the first species of the first n_heldout genera also gets a HELD-OUT sibling (world.heldout: (parent strain id, sequence)):
    a new species of an indexed genus that is NOT in the index -- the parent's genome with heldout_div substitutions, and every conserved
    segment the parent carries drawn again from the pool with its own synonymous redraws.  Reads of such an organism meet the long
    candidate runs WITHOUT an equal target in them (the ordinary metagenomic case: the exact-match shortcut of the join does not apply).
    The held-out genomes come from a generator of their own, so the indexed world is the same with and without them."""
    from metabuli_amd import synth
    g = torch.Generator(device=dev); g.manual_seed(seed)
    g2 = torch.Generator(device=dev); g2.manual_seed(seed + 991)
    heldout = []
    tax = synth.Taxonomy()
    tax.add(1, 1, "no rank", "root"); tax.add(2, 1, "superkingdom", "Bacteria"); tax.add(3, 1, "superkingdom", "Eukaryota")
    nxt = 4
    n_genera = max(1, n_species // 4)
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)

    def mutate(x, rate):
        m = torch.rand(x.shape, generator=g, device=dev) < rate
        sub = (x + torch.randint(1, 4, x.shape, generator=g, device=dev, dtype=torch.uint8)) & 3       # a DIFFERENT base
        return torch.where(m, sub, x)
    n_slots = genome_len // seg_len
    n_seg = sum(n for _, n in CONSERVED_CLASSES)
    n_cod = seg_len // 3
    conserved = conserved and n_slots >= 3 * n_seg
    if conserved:
        syn, n_syn, sense = _codon_tables(torch, dev)
        pool = sense[torch.randint(0, len(sense), (n_seg, n_cod), generator=g, device=dev)]           # codons of the pool's segments
        prev = torch.tensor([f for f, n in CONSERVED_CLASSES for _ in range(n)], device=dev)
    genomes, species = [], []
    for gi in range(n_genera):
        dom = 3 if gi == n_genera - 1 else 2
        gid = nxt; nxt += 1
        tax.add(gid, dom, "genus", f"Genus{gi}")
        anc = torch.randint(0, 4, (genome_len,), generator=g, device=dev, dtype=torch.uint8)
        for sidx in range(4):
            sid = nxt; nxt += 1
            tax.add(sid, gid, "species", f"Genus{gi} species{sidx}")
            species.append(sid)
            tid = nxt; nxt += 1
            tax.add(tid, sid, "no rank", f"Genus{gi} species{sidx} strain0")
            seq = mutate(mutate(anc, 0.15), 0.01)
            if conserved:
                carried = torch.nonzero(torch.rand(n_seg, generator=g, device=dev) < prev).flatten()
                if len(carried):
                    cod = pool[carried]
                    u = torch.rand(cod.shape, generator=g, device=dev)
                    r = torch.randint(0, 1 << 30, cod.shape, generator=g, device=dev)
                    cod = torch.where(u < p_syn, syn[cod, r % n_syn[cod]], cod)
                    cod = torch.where((u >= p_syn) & (u < p_syn + p_nonsyn), sense[r % len(sense)], cod)
                    nt = torch.stack(((cod >> 4) & 3, (cod >> 2) & 3, cod & 3), dim=2).reshape(len(carried), n_cod * 3).to(torch.uint8)
                    slots = torch.randperm(n_slots, generator=g, device=dev)[: len(carried)]
                    seq[: n_slots * seg_len].view(n_slots, seg_len)[slots, : n_cod * 3] = nt
            genomes.append((tid, acgt[seq.long()].cpu().numpy()))
            if sidx == 0 and len(heldout) < n_heldout:
                hm = torch.rand(seq.shape, generator=g2, device=dev) < heldout_div
                hs = torch.where(hm, (seq + torch.randint(1, 4, seq.shape, generator=g2, device=dev, dtype=torch.uint8)) & 3, seq)
                if conserved and len(carried):
                    cod = pool[carried]
                    u = torch.rand(cod.shape, generator=g2, device=dev)
                    r = torch.randint(0, 1 << 30, cod.shape, generator=g2, device=dev)
                    cod = torch.where(u < p_syn, syn[cod, r % n_syn[cod]], cod)
                    cod = torch.where((u >= p_syn) & (u < p_syn + p_nonsyn), sense[r % len(sense)], cod)
                    nt = torch.stack(((cod >> 4) & 3, (cod >> 2) & 3, cod & 3), dim=2).reshape(len(carried), n_cod * 3).to(torch.uint8)
                    hs[: n_slots * seg_len].view(n_slots, seg_len)[slots, : n_cod * 3] = nt
                heldout.append((tid, acgt[hs.long()].cpu().numpy()))
    lo = nxt
    for i in range(n_filler_species):
        tax.add(nxt, 2, "species", f"filler{i}"); nxt += 1
    w = synth.World(tax, genomes, species, lo, nxt - 1)
    w.heldout = heldout
    return w


def _mix64(torch, x):
    """splitmix64 finaliser on int64 tensors (wrapping arithmetic)"""
    x = (x ^ (x >> 30).bitwise_and(0x3FFFFFFFF)) * -4658895280553007687          # 0xBF58476D1CE4E5B9
    x = (x ^ (x >> 27).bitwise_and(0x1FFFFFFFFF)) * -7723592293110705685         # 0x94D049BB133111EB
    return x ^ (x >> 31).bitwise_and(0x1FFFFFFFF)


# shared-run multiplier by a hash of the amino-acid part: most shared runs stay as the genomes give them, a few grow 2 - 8 x
HOT_MULT = (0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 3, 7)


def hot_run_extras(torch, dev, hv, tax_lo, tax_span, hot_min, seed):
    """Further species for the candidate runs that the genomes already share: hv = the genome-derived target values of one sign half,
    ascending.  A run (equal amino-acid part) of r >= hot_min entries gets m x r extra entries, m = HOT_MULT[hash of the amino-acid
    part] (the longest runs of ~1900 genome-derived entries reach ~15 000): each extra is the DNA of one entry of its run with one to
    three codons redrawn among the OTHER codons the index holds for that amino acid (the letter -> codon-id table is read off the
    genome-derived entries themselves), filed under a species of the filler range.  Returns (values, taxids) ordered by (value, taxid)."""
    aa = hv >> 24
    uniq, counts = torch.unique_consecutive(aa, return_counts=True)
    starts = torch.cumsum(counts, 0) - counts
    mult = torch.tensor(HOT_MULT, device=dev)[(_mix64(torch, uniq + seed) >> 7) & 15]
    hot = (counts >= hot_min) & (mult > 0)
    if not bool(hot.any()):
        return hv[:0], torch.empty(0, dtype=torch.int32, device=dev)
    h_start, h_cnt, h_aa, mult = starts[hot], counts[hot], uniq[hot], mult[hot]
    del uniq, counts, starts, aa, hot
    # valid codon ids of every 5-bit amino-acid letter, from a sample of the entries (letter j of 8: bits 24 + 5 (7 - j), its codon id: bits 3 (7 - j))
    smp = hv[:: max(1, len(hv) // 4_000_000)]
    tbl = torch.zeros((32, 8), dtype=torch.bool, device=dev)
    for j in range(8):
        tbl[(smp >> (24 + 5 * (7 - j))) & 31, (smp >> (3 * (7 - j))) & 7] = True
    n_cid = tbl.sum(1)
    cid_list = torch.zeros((32, 8), dtype=torch.int64, device=dev); cid_pos = torch.zeros((32, 8), dtype=torch.int64, device=dev)
    for L in range(32):
        ids = torch.nonzero(tbl[L]).flatten()
        cid_list[L, : len(ids)] = ids; cid_pos[L, ids] = torch.arange(len(ids), device=dev)
    n_extra = h_cnt * mult
    tot = int(n_extra.sum().item())
    run = torch.repeat_interleave(torch.arange(len(h_cnt), device=dev), n_extra)
    k = torch.arange(tot, device=dev) - (torch.cumsum(n_extra, 0) - n_extra)[run]
    e_aa = h_aa[run]
    h = _mix64(torch, e_aa * 1000003 + k + seed)
    dna = hv[h_start[run] + ((h >> 3) & 0x7FFFFFFF) % h_cnt[run]] & 0xFFFFFF
    n_mut = 1 + ((h >> 40) & 0xFFFF) % 3
    for m in range(3):
        hm = _mix64(torch, h + 77 * (m + 1))
        sh = 3 * (7 - ((hm >> 9) & 7))                               # the codon that changes
        L = (e_aa >> (5 * (7 - ((hm >> 9) & 7)))) & 31
        cur = (dna >> sh) & 7
        n = n_cid[L]
        other = (cid_pos[L, cur] + 1 + ((hm >> 20) & 0xFFFF) % torch.clamp(n - 1, min=1)) % torch.clamp(n, min=1)
        newc = torch.where(n > 1, cid_list[L, other], cur)
        dna = torch.where(m < n_mut, (dna & ~(7 << sh)) | (newc << sh), dna)
    ev = (e_aa << 24) | dna
    et = (tax_lo + ((h >> 20) & 0x7FFFFFFF) % tax_span).to(torch.int32)
    o = torch.sort(et, stable=True).indices
    ev, et = ev[o], et[o]
    o = torch.sort(ev, stable=True).indices
    return ev[o], et[o]


def extract_targets(ctx, M, world, params, torch=None, dev=None, genomes_per_call=48, hot_min=0, seed=0):
    """Six-frame (sync)metamers of every genome, on the GPU, via the public extraction entry point (long-read geometry,
    overlapping 20 kb pieces), a few dozen genomes per call; per-genome de-duplication and the final (value, taxid) order on the
    device when torch is given (thousands of genomes), else numpy.  hot_min > 0 (device path): runs shared by at least that many
    genome-derived entries are multiplied by further species (hot_run_extras).  Returns (values, taxids, n_extras)."""
    piece, ov = 20000, 32
    p = M.default_params(seq_mode=3, syncmer=params.syncmer, smer_len=params.smer_len)
    vals, tids = [], []
    if torch is not None:
        qcuts = torch.tensor([-2**62, 0, 2**62], dtype=torch.int64, device=dev)
    for g0 in range(0, len(world.genomes), genomes_per_call):
        chunk = world.genomes[g0:g0 + genomes_per_call]
        seqs, owner = [], []
        for gi, (tid, g) in enumerate(chunk):
            for st in range(0, len(g), piece - ov):
                seqs.append(g[st:st + piece])
                owner.append(gi)
                if st + piece >= len(g):
                    break
        offs = np.zeros(len(seqs) + 1, np.uint64)
        offs[1:] = np.cumsum([len(x) for x in seqs])
        bases = np.concatenate(seqs).astype(np.uint8)
        k, _, _ = ctx.extract(p, bases, offs)
        owner = np.asarray(owner, dtype=np.int64)
        seq = ((k["qinfo"] >> np.uint64(32)) & np.uint64(0x1FFFFFFF)).astype(np.int64) - 1       # emission order = piece order = genome order
        first = np.searchsorted(owner[seq], np.arange(len(chunk) + 1))
        if torch is not None:
            kv = torch.from_numpy(np.ascontiguousarray(k["value"]).view(np.int64)).to(dev)
        for gi, (tid, g) in enumerate(chunk):
            if torch is not None:
                v = torch.unique(kv[first[gi]:first[gi + 1]])                                   # signed ascending = unsigned quarters [2, 3, 0, 1] of the value range
                c = torch.searchsorted(v, qcuts).tolist()
                vals.append((v[c[1]:c[2]], v[c[2]:], v[:c[0]], v[c[0]:c[1]])); tids.append(tid)
            else:
                v = np.unique(k["value"][first[gi]:first[gi + 1]])
                vals.append(v); tids.append(np.full(len(v), tid, np.int32))
    # one strain per species here, so (value, species) pairs are already unique; strain ids rise with the genome order,
    # so a STABLE sort by value of the genome-major concatenation is (value, taxid) order
    if torch is not None:
        # unsigned 64-bit order = the four quarters of the value range one after another, each ascending as int64 (all of one sign);
        # the quarters are sorted separately (torch.sort and boolean masks take < 2^31 elements per call; with 2400 genomes the
        # non-negative values alone are ~2 G)
        out_v, out_t, n_extras = [], [], 0
        for half in (0, 1, 2, 3):
            hv = torch.cat([x[half] for x in vals])
            ht = torch.cat([torch.full((len(x[half]),), tid, dtype=torch.int32, device=dev) for x, tid in zip(vals, tids)])
            if len(hv) >= 2**31:
                raise SystemExit(f"{len(hv)} genome-derived metamers in one quarter of the value range: more than one torch.sort call takes")
            if len(hv) == 0:
                continue
            order = torch.sort(hv, stable=True).indices
            hv, ht = hv[order], ht[order]
            del order
            if hot_min > 0 and len(hv):
                ev, et = hot_run_extras(torch, dev, hv, world.filler_tax_lo, world.filler_tax_hi - world.filler_tax_lo + 1, hot_min, seed)
                if len(ev):
                    if len(hv) + len(ev) >= 2**31:
                        raise SystemExit(f"genome-derived metamers ({len(hv)}) + shared-run extras ({len(ev)}) of one quarter of the value range exceed one torch.sort call")
                    n_extras += len(ev)
                    # filler species ids lie above every strain id: behind the genome-derived entries of the same value, a stable sort by value keeps (value, taxid) order
                    hv = torch.cat([hv, ev]); ht = torch.cat([ht, et])
                    del ev, et
                    order = torch.sort(hv, stable=True).indices
                    hv, ht = hv[order], ht[order]
                    del order
            out_v.append(hv.cpu().numpy().view(np.uint64)); out_t.append(ht.cpu().numpy())
            del hv, ht
        del vals
        torch.cuda.empty_cache()
        return np.concatenate(out_v), np.concatenate(out_t), n_extras
    vals = np.concatenate(vals); tids = np.concatenate(tids)
    order = np.lexsort((tids, vals))
    return vals[order], tids[order], 0


def gen_reads(torch, dev, genomes, n_reads, read_len, frac_random, err, seed, paired=False, frag_len=400):
    """Reads sampled from the genomes (both strands, substitutions) + random
    reads, generated on the device so that the inputs are HBM-resident.
    paired: fragments of frag_len bases, mate 1 = its first read_len bases, mate 2 = the first read_len bases of its
    reverse complement (BASELINE.json configs[3] shape); returns (bases1, offs, bases2)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    G = torch.from_numpy(np.concatenate([x for _, x in genomes]).astype(np.uint8)).to(dev)
    lens = torch.tensor([len(x) for _, x in genomes], device=dev, dtype=torch.int64)
    starts = torch.cumsum(lens, 0) - lens
    comp = torch.full((256,), ord("N"), dtype=torch.uint8, device=dev)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    out = torch.empty(n_reads * read_len, dtype=torch.uint8, device=dev)
    out2 = torch.empty(n_reads * read_len, dtype=torch.uint8, device=dev) if paired else None
    span = frag_len if paired else read_len
    ar = torch.arange(span, device=dev, dtype=torch.int64)
    chunk = max(1, min(1_000_000, 200_000_000 // span))
    for c0 in range(0, n_reads, chunk):
        n = min(chunk, n_reads - c0)
        gi = torch.randint(0, len(lens), (n,), generator=g, device=dev)
        st = (torch.rand(n, generator=g, device=dev) * (lens[gi] - span + 1).float()).long().clamp_(min=0) + starts[gi]
        r = G[st[:, None] + ar[None, :]]
        rc = torch.rand(n, generator=g, device=dev) < 0.5
        r = torch.where(rc[:, None], comp[r.long()].flip(1), r)
        sub = torch.rand(n, span, generator=g, device=dev) < err
        r = torch.where(sub, acgt[torch.randint(0, 4, (n, span), generator=g, device=dev)], r)
        rnd = torch.rand(n, generator=g, device=dev) < frac_random
        r = torch.where(rnd[:, None], acgt[torch.randint(0, 4, (n, span), generator=g, device=dev)], r)
        if paired:
            out[c0 * read_len:(c0 + n) * read_len] = r[:, :read_len].reshape(-1)
            out2[c0 * read_len:(c0 + n) * read_len] = comp[r.long()].flip(1)[:, :read_len].reshape(-1)
        else:
            out[c0 * read_len:(c0 + n) * read_len] = r.reshape(-1)
    offs = torch.arange(n_reads + 1, device=dev, dtype=torch.int64) * read_len
    if paired:
        return out, offs, out2
    return out, offs


# --------------------------------------------------------------------------------------------------------------------
# parity: a sample of a batch against the oracle on a sub-database of the timed index
# --------------------------------------------------------------------------------------------------------------------
def _unsigned_split(d_values, T):
    """the flat array is sorted as UNSIGNED 64-bit; as int64 it is [non-negative ascending | negative ascending]: the split"""
    lo, hi = 0, T
    while lo < hi:
        mid = (lo + hi) // 2
        if int(d_values[mid].item()) >= 0:
            lo = mid + 1
        else:
            hi = mid
    return lo


def closure_positions(torch, d_values, T, q_values, stride=0):
    """Positions (ascending, device tensor) of every target of the flat device index whose amino-acid part equals that of a query
    metamer, plus the index's true last entry, plus -- stride > 0 -- every stride-th target: computed with torch.searchsorted on the
    flat array, i.e. by nothing of the library under test.  Matches(q) depends on no other target (C(q) = {t < T-1 :
    AA(t) = AA(q)}), so the oracle on any sub-database that holds these positions gives every read of the sample the answer the GPU
    gives against the whole index."""
    dev = d_values.device
    aa = torch.unique(torch.from_numpy(np.ascontiguousarray(q_values).view(np.int64)).to(dev) >> 24)       # arithmetic shift: sign-extended, still one key per amino-acid part
    v = d_values[:T]
    n_pos = _unsigned_split(d_values, T)
    lo_key = aa << 24
    hi_key = lo_key | 0xFFFFFF
    neg = lo_key < 0
    pieces = []
    for sel, base, seg in ((~neg, 0, v[:n_pos]), (neg, n_pos, v[n_pos:])):
        if not bool(sel.any()) or len(seg) == 0:
            continue
        a = torch.searchsorted(seg, lo_key[sel], right=False)
        b = torch.searchsorted(seg, hi_key[sel], right=True)
        cnt = b - a
        keep = cnt > 0
        a, cnt = a[keep], cnt[keep]
        tot = int(cnt.sum().item())
        if tot:
            first = torch.cumsum(cnt, 0) - cnt
            pieces.append(torch.repeat_interleave(a - first, cnt) + torch.arange(tot, device=dev) + base)
    pieces.append(torch.tensor([T - 1], device=dev))           # the entry the `t < T-1` rule excludes must be the sub-database's last one too
    pos_c = torch.unique(torch.cat(pieces))
    del pieces
    if stride <= 0:
        return pos_c
    # union with the strided positions without sorting 10^9 numbers: both lists are ascending, so every element's place in the
    # union is its own rank plus its rank in the other list
    pos_c = pos_c[pos_c % stride != 0]
    n_c, n_s = len(pos_c), (T + stride - 1) // stride
    out = torch.empty(n_c + n_s, dtype=torch.int64, device=dev)
    out[torch.arange(n_c, device=dev) + torch.div(pos_c, stride, rounding_mode="floor") + 1] = pos_c
    for j0 in range(0, n_s, 1 << 27):
        j = torch.arange(j0, min(n_s, j0 + (1 << 27)), device=dev)
        out[j + torch.searchsorted(pos_c, j * stride)] = j * stride
    return out


def gather_subindex(d_values, d_info, pos):
    cv = d_values[pos].cpu().numpy().view(np.uint64)
    ct = (d_info[pos].cpu().numpy().view(np.int32) & np.int32(0x7FFFFFFF))
    assert (cv[1:] >= cv[:-1]).all()
    return cv, ct


def sample_closure(ctx, torch, params, d_values, d_info, T, d_bases, d_bases2, read_len, n, stride=0):
    """the candidate closure (+ optional strided sample) of the first n reads of a batch, taken from the flat arrays"""
    sb = d_bases[: n * read_len].cpu().numpy()
    so = np.arange(n + 1, dtype=np.uint64) * np.uint64(read_len)
    sk, _, _ = ctx.extract(params, sb, so, d_bases2[: n * read_len].cpu().numpy() if d_bases2 is not None else None, so if d_bases2 is not None else None)
    pos = closure_positions(torch, d_values, T, sk["value"], stride)
    n_k = len(sk)
    del sk
    cv, ct = gather_subindex(d_values, d_info, pos)
    return dict(values=cv, taxids=ct, n_reads=n, n_kmers=n_k, stride=stride)


def compare_with_oracle(M, res, tt, tc, R):
    """GPU per-read results (compacted taxcnt lists) against the oracle's answer R for the same reads: taxon, classified
    flag, score on the fp32 bit pattern, query lengths, and the taxID:match_count lists.  Reads the oracle flags as
    std::sort-ambiguous are excluded and counted.  Returns a dict for the JSON line."""
    ro = R["results"]
    amb = ro["flag"] != 0
    bad = (res["classification"] != ro["classification"]) | (res["is_classified"] != ro["is_classified"]) | \
          (res["score"].view(np.uint32) != ro["score"].view(np.uint32)) | (res["qlen"] != ro["qlen"]) | (res["qlen2"] != ro["qlen2"]) | \
          (res["n_taxcnt"] != ro["n_taxcnt"])
    bad &= ~amb
    # taxID:match_count lists of the reads that agree so far (equal lengths there): gather both sides by their offsets
    ok = np.flatnonzero(~amb & ~bad)
    n = ro["n_taxcnt"][ok].astype(np.int64)
    tot = int(n.sum())
    first = np.zeros(len(ok) + 1, np.int64); np.cumsum(n, out=first[1:])
    within = np.arange(tot, dtype=np.int64) - np.repeat(first[:-1], n)
    ig = np.repeat(res["taxcnt_off"][ok].astype(np.int64), n) + within
    io = np.repeat(ro["taxcnt_off"][ok].astype(np.int64), n) + within
    diff = (tt[ig] != R["tc_tax"][io]) | (tc[ig] != R["tc_cnt"][io])
    list_bad = int(len(np.unique(np.repeat(np.arange(len(ok)), n)[diff])))
    return dict(reads=int(len(ro)), mismatches=int(bad.sum()) + list_bad, ambiguous_excluded=int(amb.sum()),
                classified=int((ro["is_classified"] != 0).sum()),
                checked="classification, is_classified, score (fp32 bits), query lengths, taxID:match_count lists")


def gpu_sample(ctx, M, torch, dev, index, params, d_bases, d_bases2, read_len, n):
    """the first n reads of a device-resident batch through the benchmarked entry point against `index`"""
    paired = params.seq_mode == 2
    offs = torch.arange(n + 1, device=dev, dtype=torch.int64) * read_len
    s_res = torch.empty(n * 24, dtype=torch.uint8, device=dev)
    s_cap = n * (20 + read_len // 9) * (2 if paired else 1) + 1024
    s_tt = torch.empty(s_cap, dtype=torch.int32, device=dev); s_tc = torch.empty(s_cap, dtype=torch.int32, device=dev)
    ntc = ctx.classify_batch_device(index, params, d_bases.data_ptr(), offs.data_ptr(), d_bases2.data_ptr() if paired else 0,
                                    offs.data_ptr() if paired else 0, n, n * read_len * (2 if paired else 1),
                                    s_res.data_ptr(), s_tt.data_ptr(), s_tc.data_ptr(), s_cap)
    torch.cuda.synchronize()
    res = np.frombuffer(s_res.cpu().numpy().tobytes(), dtype=M.result_dt)
    return M.compact_taxcnt(res, s_tt[:ntc].cpu().numpy(), s_tc[:ntc].cpu().numpy().view(np.uint32)) + (int(ctx.last_stats().n_matches),)


def oracle_parity(ctx, M, torch, dev, index, params, taxdir, d_bases, d_bases2, read_len, sub, T, time_cpu=False, label=""):
    """(a) The oracle (CPU restatement of the reference algorithm, test infrastructure) on the sample's reads against the
    sub-database `sub` of the timed index (sample_closure: the candidate closure of the sample, for the headline sample also every
    stride-th target); (b) the SAME reads through the benchmarked entry point against THE TIMED INDEX ITSELF, compared read by read.
    time_cpu: the oracle run is also the reported CPU baseline (all host cores, one core on a part of the sample, cold first run).
    Outside the timed region."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import Oracle, default_params as odp
    orc = Oracle()
    cv, ct, n = sub["values"], sub["taxids"], sub["n_reads"]
    d = tempfile.mkdtemp(prefix="mtb_sub_")
    op = odp(seq_mode=params.seq_mode, syncmer=params.syncmer, smer_len=params.smer_len)
    t0 = time.perf_counter()
    orc.write_db(d, cv, ct, op)
    t_write = time.perf_counter() - t0
    tax = orc.load_taxonomy(taxdir)
    db = orc.open_db(d, tax, op)
    paired = params.seq_mode == 2
    bases = d_bases[: n * read_len].cpu().numpy()
    offs = np.arange(n + 1, dtype=np.uint64) * np.uint64(read_len)
    bases2 = d_bases2[: n * read_len].cpu().numpy() if paired else None
    offs2 = offs if paired else None
    ncores = os.cpu_count() or 1
    cpu = None
    if time_cpu:
        cold_s = None               # (no cold-cache run: the GPU pool does not let a job drop the machine's page cache; the files were just written, so every run is warm)
        nw = min(n, 20000)          # untimed: creates the OpenMP thread pool
        orc.classify_batch(db, tax, op, bases[: nw * read_len], offs[: nw + 1], bases2[: nw * read_len] if paired else None, offs[: nw + 1] if paired else None, threads=ncores)
    t0 = time.perf_counter()
    R = orc.classify_batch(db, tax, op, bases, offs, bases2, offs2, threads=ncores)
    dt = time.perf_counter() - t0
    stage_s = dict(orc.last_stage_s); oracle_counts = dict(orc.last_counts)
    if time_cpu:
        n1 = max(1, min(n // 8, 50000))          # the single-thread figure on a part of the sample (same code, threads=1)
        t1 = time.perf_counter()
        orc.classify_batch(db, tax, op, bases[: n1 * read_len], offs[: n1 + 1], bases2[: n1 * read_len] if paired else None, offs[: n1 + 1] if paired else None, threads=1)
        dt1 = time.perf_counter() - t1
        cls = int((R["results"]["is_classified"] != 0).sum())
        fbytes = int(sum(os.path.getsize(os.path.join(d, f)) for f in ("diffIdx", "info")))
        cpu = dict(value=n / dt / 1e6, unit="Mreads/s", cores=ncores, kind="port", cpu_model=cpu_model(),
                   single_thread_value=n1 / dt1 / 1e6, stage_seconds={k: round(v, 3) for k, v in stage_s.items()},
                   cold_cache_first_run_s=cold_s, cold_cache_note="warm runs only: the database files were written just before and sit in the page cache",
                   numa=numa_layout(), index_targets=int(len(cv)), index_file_bytes=fbytes, timed_index_targets=int(T),
                   sample=f"first {n} x {'2 x ' if paired else ''}{read_len} bp reads of the timed batch vs a {len(cv)}-target sub-database of the timed index "
                          f"(every {sub['stride']}th target + the candidate closure of the sample: same answers as against all {T} targets); "
                          f"oracle/liboracle.so, OpenMP, {ncores} threads, {dt:.1f} s (match stage {stage_s.get('match', 0):.1f} s; 1 thread on {n1} reads: {dt1:.1f} s)")
    dead = None
    if time_cpu:
        # VERDICT r3 item 6(i), the measurement: how many matches can never score?  A match takes part in a path only inside a
        # (species, frame) group of >= 2 matches (Taxonomer.cpp:342); a species none of whose groups in the read has two matches gets no
        # path, hence no score, hence is never the read's species: all its matches in that read are dead weight for the join's stores.
        n2 = min(n, 50000)
        S = orc.classify(db, tax, op, bases[: n2 * read_len], offs[: n2 + 1], bases2[: n2 * read_len] if paired else None, offs[: n2 + 1] if paired else None, threads=ncores)
        mm = S["matches"]
        if len(mm):
            seq = ((mm["qinfo"] >> np.uint64(32)) & np.uint64(0x1FFFFFFF)).astype(np.int64)
            frame = (mm["qinfo"] >> np.uint64(61)).astype(np.int64)
            g_key = (seq * (1 << 32) + mm["species_id"].astype(np.int64)) * 8 + frame
            _, g_inv, g_cnt = np.unique(g_key, return_inverse=True, return_counts=True)
            s_key = seq * (1 << 32) + mm["species_id"].astype(np.int64)
            _, s_inv = np.unique(s_key, return_inverse=True)
            best = np.zeros(s_inv.max() + 1, np.int64)
            np.maximum.at(best, s_inv, g_cnt[g_inv])
            dead = dict(sample_reads=int(n2), matches=int(len(mm)), in_groups_of_one=int((g_cnt[g_inv] == 1).sum()),
                        of_species_without_any_pair=int((best[s_inv] == 1).sum()),
                        note="matches of a (read, species) none of whose (species, frame) groups holds two matches can never score (Taxonomer.cpp:342): "
                             "the share of the join's slot stores a per-read sketch could skip")
            dead["dead_fraction"] = dead["of_species_without_any_pair"] / max(1, dead["matches"])
        del S, mm
    g_res, g_tt, g_tc, g_matches = gpu_sample(ctx, M, torch, dev, index, params, d_bases, d_bases2, read_len, n)
    par = compare_with_oracle(M, g_res, g_tt, g_tc, R)
    par["matches"] = g_matches; par["oracle_matches"] = int(oracle_counts["matches"])
    if par["matches"] != par["oracle_matches"]:
        par["mismatches"] += 1
    stt = index.state()
    par["index"] = (f"the timed index itself: {T} targets, directory depth {stt['dir_depth']}, {'packed 8-byte words' if stt['packed'] else 'flat {value, info}'}"
                    f"{', sealed' if stt['sealed'] else ''}; oracle on a sub-database of {len(cv)} targets (candidate closure of the sample's metamers"
                    f"{', every %dth target' % sub['stride'] if sub['stride'] else ''} and the index's last entry, gathered with torch.searchsorted from the flat arrays before packing)")
    par["sub_database_targets"] = int(len(cv)); par["oracle_seconds"] = round(dt, 2); par["db_write_seconds"] = round(t_write, 2)
    if dead is not None:
        par["dead_matches"] = dead
    log(f"[rank 0] parity {label}: {par['reads']} reads, {par['mismatches']} mismatches, {par['matches']} matches (oracle {par['oracle_matches']}), sub-database {len(cv)} targets, oracle {dt:.1f}s")
    return cpu, par


def numa_layout():
    """NUMA nodes of the host and their CPU lists (printed next to the CPU baseline)"""
    out = {}
    try:
        base = "/sys/devices/system/node"
        for d in sorted(os.listdir(base)):
            if d.startswith("node") and d[4:].isdigit():
                out[d] = open(os.path.join(base, d, "cpulist")).read().strip()
    except OSError:
        pass
    return out


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def silence_other_ranks(rank):
    """torch.distributed.run interleaves every rank's stdout: only rank 0 may write there (C libraries included)"""
    if rank != 0:
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)


def finish(dist, line):
    """Rank 0's JSON line is the LAST thing on stdout: collectives are torn down first and whatever C libraries (RCCL prints its
    library path through C stdio, which is block-buffered on a pipe) left in the C-level buffer is flushed before it."""
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    if line is not None:
        print(line, flush=True)


# --------------------------------------------------------------------------------------------------------------------
# per-kernel roofline of one profiled step
# --------------------------------------------------------------------------------------------------------------------
def hist_summary(by_log2, weights=None):
    """quantiles (upper edge of the log2 bin) of a run-length histogram"""
    c = np.asarray(by_log2, dtype=np.float64)
    tot = c.sum()
    if tot == 0:
        return dict(n=0)
    cum = np.cumsum(c) / tot
    q = {f"p{int(p * 100) if p < 0.995 else 99.9}": int(2 ** (int(np.searchsorted(cum, p)) + 1) - 1) for p in (0.5, 0.9, 0.99, 0.999)}
    q["max_bin_upper"] = int(2 ** (int(np.flatnonzero(c)[-1]) + 1) - 1)
    q["n"] = int(tot)
    return q


def pmc_traffic(key, workload_tuple, kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes of this very workload (profiles/pmc_traffic_<key>.json), or None"""
    for name in (f"pmc_traffic_{key}.json", "pmc_traffic.json"):
        try:
            pj = json.load(open(os.path.join(ROOT, "profiles", name)))
            wl = pj["workload"]
            if (wl["reads"], wl["read_len"], wl["targets"], wl["seq_mode"]) == workload_tuple and wl.get("key", "default") == key and kernel in pj["kernels"]:
                kk = pj["kernels"][kernel]
                return (2.0 * kk.get("fetch_size_kb", 0.0) + kk.get("write_size_kb", 0.0)) * 1024.0, f"{pj['source']}: {pj['correction']}"
        except (OSError, KeyError, ValueError):
            continue
    return None, f"no PMC passes for this workload (profiles/pmc_traffic_{key}.json)"


def profiled_step(ctx, M, index, params, step_fn, streams, key, workload_tuple):
    """one extra, untimed, profiled step on ONE stream (kernels not overlapped, full-batch launches): HIP events on the library's
    stream around every kernel launch -> per-kernel ms, the contract roofline of the dominant kernel (SURVEY 8(d) bytes), its
    PMC traffic when the passes exist, and for the directory join the index-side working set (`footprint`) with
    frac_design = (least_fetch_bytes + 16 B x matches) / launch time / peak: the bandwidth fraction against what THIS design has
    to move at least."""
    ctx.set_streams(1)
    ctx.set_profiling(True)
    step_fn()
    ps = ctx.last_stats()
    ctx.set_profiling(False)
    ctx.set_streams(streams)
    kern = {M.KERNEL_NAMES[i]: dict(ms=float(ps.ms_kernel[i]), launches=int(ps.n_launch[i])) for i in range(len(M.KERNEL_NAMES))}
    footprint, runs = None, None
    try:
        fp = ctx.join_footprint(index)
        least = fp.target_sectors * 64 + fp.dir_sectors * 64 + 16 * fp.n_queries
        footprint = dict(query_metamers=int(fp.n_queries), distinct_buckets=int(fp.distinct_buckets), buckets=int(fp.n_buckets),
                         directory_sectors_64B=int(fp.dir_sectors), target_sectors_64B=int(fp.target_sectors),
                         target_sectors_total=int(fp.n_targets * 8 // 64), target_fraction_touched=fp.target_sectors * 64 / max(1, fp.n_targets * 8),
                         least_fetch_bytes=int(least),
                         note="distinct 64-byte sectors the batch's queries address (bucket spans of the target array, directory words) + 16 B per query: "
                              "the least k_join_dir can fetch for this batch; 12 x T of the contract formula is a streaming-merge figure this kernel never pays")
        rh = ctx.join_run_histogram(index)
        runs = dict(queries_without_candidate=rh["no_candidate"], queries_by_log2_run_length=rh["queries_by_log2"][:16],
                    run_targets_by_log2_run_length=rh["candidates_by_log2"][:16], quantiles=hist_summary(rh["queries_by_log2"]),
                    run_targets_met=int(sum(rh["candidates_by_log2"])),
                    queries_with_own_dna_in_a_long_run=rh["exact_queries"], run_targets_not_scanned_for_them=rh["exact_run_targets"],
                    note="length of the candidate run (targets sharing the query's amino-acid part) every query metamer of the step meets; bin b = lengths 2^b .. 2^(b+1)-1")
    except M.MtbError as e:
        log(f"no join footprint / run histogram: {e}")
    Kq, Mm, N, L = ps.n_kmers, ps.n_matches, ps.n_reads, ps.n_bases
    # algorithmic bytes of one whole step per kernel (SURVEY.md 8(d) per-stage split);
    # a step launches every kernel once per stream (and per radix pass): bytes per launch = total / launches
    alg_step = {"extract_count": L, "extract_emit": L + 16 * Kq, "radix_hist": 16 * Kq, "radix_scatter": 32 * Kq,
                "join": 16 * Kq + 24 * Mm, "regroup": 48 * Mm, "segsort": 48 * Mm, "score": 24 * Mm + 16 * N, "score_fast": 24 * Mm + 16 * N}
    if params.kmer_format == 2 and kern["radix_hist"]["launches"]:
        # the fused path's histograms read the 2-byte digit side arrays and write the 4-byte tile table, not the 16-byte records
        alg_step["radix_hist"] = kern["radix_hist"]["launches"] * (2 * Kq + 4 * 512 * ((Kq + 4095) // 4096))
    alg = {k: v / max(1, kern[k]["launches"]) for k, v in alg_step.items()}
    if kern.get("score_fast", {}).get("launches") and params.seq_mode != 3:         # the two scoring kernels share the reads
        gfrac = ps.n_generic_reads / max(1, N)
        alg["score"] *= gfrac; alg["score_fast"] *= 1.0 - gfrac
    # the 12*T term of SURVEY 8(d) is charged only to a launch that really streams the index: one whose queries address at least half of
    # the target array's 64-byte sectors (the 10 M-read headline: 0.84).  A leg of 2 M reads or a long-read sub-batch walks a fraction of it:
    # such a launch is charged the sectors it addresses (directory + target spans) instead, so that no `frac` is inflated by bytes the launch
    # never had to move (1.45 was printed for the 20 k long-read leg)
    streams_index = footprint is not None and footprint["target_fraction_touched"] >= 0.5 and kern["join"]["launches"] <= 1
    index_term = 12 * ps.n_targets if streams_index else ((footprint["target_sectors_64B"] + footprint["directory_sectors_64B"]) * 64 if footprint is not None else 0)
    alg["join"] += index_term
    dom = max((k for k in alg), key=lambda k: kern[k]["ms"])
    avg_ms = kern[dom]["ms"] / max(1, kern[dom]["launches"])
    achieved = alg[dom] / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    traffic, traffic_note = pmc_traffic(key, workload_tuple, dom)
    roofline_all = {k: dict(ms=round(kern[k]["ms"], 3), launches=kern[k]["launches"], algorithmic_gb_per_launch=round(alg[k] / 1e9, 3),
                            achieved_gb_s=round(alg[k] / (kern[k]["ms"] / max(1, kern[k]["launches"]) * 1e-3) / 1e9, 1),
                            frac=round(alg[k] / (kern[k]["ms"] / max(1, kern[k]["launches"]) * 1e-3) / 1e9 / PEAK_GBS, 4))
                    for k in alg if kern[k]["ms"] > 0}
    effective = (traffic / (avg_ms * 1e-3) / 1e9 / PEAK_GBS) if (traffic and avg_ms > 0) else None
    write_amp = None
    try:
        pj = json.load(open(os.path.join(ROOT, "profiles", f"pmc_traffic_{key}.json")))
        if dom == "join" and "join" in pj["kernels"] and Mm:
            write_amp = pj["kernels"]["join"]["write_size_kb"] * 1024.0 / (16.0 * Mm / max(1, kern["join"]["launches"]))
    except (OSError, KeyError, ValueError):
        pass
    frac_design = None
    join_ms = kern["join"]["ms"] / max(1, kern["join"]["launches"])
    if footprint is not None and join_ms > 0:
        frac_design = (footprint["least_fetch_bytes"] + 16 * Mm) / (join_ms * 1e-3) / 1e9 / PEAK_GBS
    frac = achieved / PEAK_GBS
    if frac > 1.0:        # a fraction above 1 is an accounting artefact, not evidence: never printed
        achieved, frac = None, None
    roofline = dict(bound="hbm", kernel=dom, achieved=achieved, peak=PEAK_GBS, unit="GB/s", frac=frac, traffic=traffic, traffic_note=traffic_note,
                    streams_index=bool(streams_index), index_bytes_charged=int(index_term),
                    effective=effective, write_amplification=write_amp,
                    write_amplification_note="join only: PMC WRITE_SIZE per launch / (16 B x matches per launch): every scattered 16-byte slot store is a 32-byte transaction",
                    effective_note="traffic / avg_launch_ms / peak: the fraction of HBM bandwidth the kernel really moves (PMC bytes, not the contract's algorithmic bytes)",
                    frac_design=frac_design,
                    frac_design_note="join only: (distinct index sectors + directory sectors + 16 B per query + 16 B per match slot) / join launch time / peak -- the least this "
                                     "design (directory lookup, scattered slot stores) can move; the kernel is bound by the NUMBER of scattered 16-byte store transactions, not by these bytes",
                    avg_launch_ms=avg_ms, launches=kern[dom]["launches"], algorithmic_bytes_per_launch=alg[dom],
                    footprint=footprint if dom == "join" else None,
                    note="per-kernel durations from HIP events around every launch of one extra step; traffic = HBM bytes per launch from the PMC passes; "
                         "`frac` follows SURVEY 8(d)'s formula (16 Kq + 12 T + 24 M for the join) and is NOT a bandwidth fraction for the directory join: see `effective`, `frac_design` and `footprint`")
    return ps, kern, roofline, roofline_all, footprint, runs


# --------------------------------------------------------------------------------------------------------------------
# the line the driver parses (<= 8 KB) and the detail file next to it
# --------------------------------------------------------------------------------------------------------------------
LINE_LIMIT = 8192
DETAIL_NAME = "bench_detail.json"


def _pick(d, keys):
    return {k: d[k] for k in keys if d is not None and k in d}


def _r(x, nd=4):
    """floats of the line rounded to `nd` significant decimals (the detail file keeps full precision)"""
    if isinstance(x, float):
        return float(f"{x:.{nd + 3}g}")
    if isinstance(x, dict):
        return {k: _r(v, nd) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, nd) for v in x]
    return x


def headline(out, detail_path):
    """The driver's line: the contract keys + config / stage_ms / roofline / cpu_baseline / parity_sample / other_configs in their short form
    (VERDICT r5 item 1).  No prose beyond `config.workload` and `cpu_baseline.sample`; histograms, footprints, notes, per-kernel tables,
    per-rank identities and the deferred-read statistics are in the detail file.  A `frac` is printed only for a launch that streams the
    index (profiled_step sets it to None otherwise) and never above 1."""
    line = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"))
    line["config"] = _pick(out["config"], ("workload", "reads_per_gpu", "read_len", "targets", "seq_mode", "gbp_per_s", "query_metamers", "matches", "classified_fraction",
                                           "parallelism", "sub_batches_per_step", "index_sealed", "index_bytes", "tuning_steps", "join_variant", "species", "index_handover"))
    line["stage_ms"] = out.get("stage_ms")
    rf = out.get("roofline")
    line["roofline"] = _pick(rf, ("bound", "kernel", "achieved", "peak", "peak_measured", "unit", "frac", "traffic", "effective", "write_amplification",
                                  "avg_launch_ms", "launches", "algorithmic_bytes_per_launch")) if rf else None
    cb = out.get("cpu_baseline")
    line["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "sample", "cpu_model", "single_thread_value", "index_targets")) if cb else None
    ps = out.get("parity_sample")
    line["parity_sample"] = _pick(ps, ("reads", "mismatches", "matches", "oracle_matches", "ambiguous_excluded")) if ps else None
    oc = {}
    legs = dict(out.get("other_configs") or {})
    if out.get("best_case"):
        legs["best_case"] = out["best_case"]
    for name, e in legs.items():
        o = _pick(e, ("reads", "read_len", "seq_mode", "ms_per_step", "mreads_per_s", "gbp_per_s", "sub_batches", "join_variant"))
        if e.get("stage_ms"):
            o["join_ms"] = e["stage_ms"].get("join"); o["score_ms"] = e["stage_ms"].get("score")
        if e.get("parity"):
            o["parity"] = _pick(e["parity"], ("reads", "mismatches"))
        oc[name] = o
    line["other_configs"] = oc or None
    line["detail"] = detail_path
    return _r(line)


def emit_lines(out):
    """writes the full record to bench_detail.json (working directory, and its gpurun_out/ when that exists: it is what travels back from a
    GPU box) and to stderr, returns the headline line; the line is cut further if it ever came near the limit (a CPU test holds it under 8 KB)"""
    full = json.dumps(out)
    here = os.getcwd()
    paths = [os.path.join(here, DETAIL_NAME)]
    if os.path.isdir(os.path.join(here, "gpurun_out")):
        paths.append(os.path.join(here, "gpurun_out", DETAIL_NAME))
    written = None
    for pth in paths:
        try:
            with open(pth, "w") as f:
                f.write(full + "\n")
            written = written or os.path.relpath(pth, here)
        except OSError as e:
            log(f"detail file {pth} not written: {e}")
    log("[rank 0] bench detail: " + full)
    line = headline(out, written)
    text = json.dumps(line)
    if len(text) >= LINE_LIMIT:          # belt and braces: drop the optional blocks, longest first, until it fits
        for k in ("other_configs", "stage_ms", "parity_sample"):
            line[k] = None
            text = json.dumps(line)
            if len(text) < LINE_LIMIT:
                break
    log(f"[rank 0] headline line: {len(text)} bytes; detail in {written}")
    return text



def timed_leg(torch, step_fn, warmup, steps):
    for _ in range(warmup):
        step_fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def copy_peak_gbs(torch, dev, nbytes=4 << 30, reps=5):
    """what a plain device-to-device copy reaches on this GPU right now (read + write bytes / time): the practical ceiling next to the
    8 TB/s data-sheet figure"""
    try:
        n = nbytes // 8
        a = torch.empty(n, dtype=torch.int64, device=dev); b = torch.empty(n, dtype=torch.int64, device=dev)
        a.fill_(1); b.copy_(a)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            b.copy_(a)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        del a, b
        torch.cuda.empty_cache()
        return 2.0 * n * 8 / (ms * 1e-3) / 1e9
    except Exception as e:       # (no timing events on the emulated build)
        log(f"copy peak not measured: {e}")
        return None


def main(device=None):
    """device: tests only (tests/hipemu/bench_emulated.py hands in torch.device("cpu") for the library build that runs on the CPU stand-in
    of the HIP runtime); the product run takes cuda:LOCAL_RANK."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU per step")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--targets", type=float, default=16e9,
                    help="TOTAL target metamers of the synthetic index (per GPU, replicated): genome-derived + shared-run extras + filler; "
                         "16 G = SURVEY 8(d)'s GTDB-scale planning size, 192 GB flat")
    ap.add_argument("--species", type=int, default=2400,
                    help="genomes the reads are drawn from (and whose metamers are in the index); >= 200 takes the device-side generator "
                         "(2400 x 1 Mbp: 0.6 x coverage by 10 M reads); 24 is the round-3 headline (62 x coverage: this design's best case)")
    ap.add_argument("--fixed-total", action="store_true", help="(accepted for compatibility: --targets always is the total now)")
    ap.add_argument("--no-conserved", action="store_true", help="device-side generator: no conserved segments, no shared-run extras (uniform candidate runs of 1-4 entries)")
    ap.add_argument("--hot-min", type=int, default=8, help="genome-derived entries a candidate run must hold to be multiplied by further species")
    ap.add_argument("--full-parity-reads", type=int, default=16384, help="reads of the parity samples of the other configurations' legs (pairs; long reads scaled by length)")
    ap.add_argument("--genome-len", type=int, default=1_000_000)
    ap.add_argument("--filler-species", type=int, default=130_000)
    ap.add_argument("--cpu-reads", type=int, default=1_000_000, help="reads of the CPU-baseline / parity sample (the first reads of rank 0's batch)")
    ap.add_argument("--cpu-stride", type=int, default=16,
                    help="the CPU baseline's database = every stride-th target of the timed index + the candidate closure of the sample (16: 1 G + closure of 16 G targets, "
                         "~15 GB of diffIdx + info on the host; the oracle streams it from every split checkpoint)")
    ap.add_argument("--cpu-targets", type=float, default=0, help="(ignored; the CPU baseline's database is a sub-database of the timed index, see --cpu-stride)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the timed CPU baseline (the parity sample still runs the oracle)")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle comparison of the benchmarked path (and the CPU baseline)")
    ap.add_argument("--no-legs", action="store_true", help="skip the legs of the other configurations (best case, pairs, long reads)")
    ap.add_argument("--leg-pairs", type=int, default=2_000_000, help="read pairs of the paired-end leg")
    ap.add_argument("--leg-long", type=int, default=20_000, help="reads of the long-read leg (x --leg-long-len bp)")
    ap.add_argument("--leg-long-len", type=int, default=10_000)
    ap.add_argument("--leg-novel", type=int, default=2_000_000, help="reads of the held-out-organism leg (species of indexed genera that are NOT in the index)")
    ap.add_argument("--heldout", type=int, default=200, help="held-out genomes the novel leg's reads are drawn from")
    ap.add_argument("--long-parity-reads", type=int, default=5000, help="long reads of the long leg's parity sample")
    ap.add_argument("--streams", type=int, default=1, help="HIP streams a batch is pipelined over inside the library")
    ap.add_argument("--seq-mode", type=int, default=1, choices=[1, 2, 3],
                    help="1 = short single-end (configs[1]); 2 = paired-end, --reads pairs of 2 x --read-len (configs[3] shape); 3 = long reads (configs[2])")
    ap.add_argument("--partitioned", action="store_true",
                    help="SURVEY 8(e) row 2: every rank owns one value range of the index; metamers and matches travel by all-to-all "
                         "(functional/perf check of that path; the default is the replicated index)")
    ap.add_argument("--no-seal", action="store_true", help="keep the flat {value, info} arrays next to the packed state (mtb_index_seal not called)")
    ap.add_argument("--ab", default="", help="A/B legs after the timed region: ';'-separated settings (NAME=VALUE, or several joined by ',') of the library's experiment switches "
                                             "(MTB_NO_SCORE_MANY=1, MTB_JOIN_VARIANT=q1w6 ...); each is timed on the headline batch (and the best-case batch) in this very process, "
                                             "on this very index and allocation")
    ap.add_argument("--reads-from", default="index", choices=["index", "heldout"],
                    help="heldout: the timed batch's reads come from the held-out genomes (species of indexed genera that are NOT in the index) -- the novel leg's workload as the "
                         "main one, for profiler runs")
    ap.add_argument("--handover", action="store_true", help="N > 1: rank 0 builds the index once and the other ranks import it through inter-process handles (mtb_index_export / mtb_index_import); "
                                                             "default: every rank synthesises its own replica -- opening another process's handle failed or hung on some boxes of the test pool (profiles/r06_notes.md)")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--dist-backend", default="nccl", help="testing only: gloo lets several ranks share one GPU (RCCL refuses duplicate devices)")
    ap.add_argument("--shared-gpu", action="store_true", help="testing only: every rank uses cuda:0")
    args = ap.parse_args()

    import torch  # before libmtb: both must share one HIP runtime (libamdhip64.so.7)
    rank = int(os.environ.get("RANK", "0")); world_size = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    silence_other_ranks(rank)
    if world_size != args.gpus:
        log(f"warning: WORLD_SIZE={world_size} but --gpus {args.gpus}")
    if args.shared_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = device if device is not None else torch.device("cuda", local_rank)
    dist = None
    if world_size > 1:
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=dev)
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world_size)
    elif args.partitioned:
        import torch.distributed as dist
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{29400 + os.getpid() % 500}", rank=0, world_size=1, device_id=dev)
    import metabuli_amd as M
    ctx = M.Context(local_rank)
    ctx.set_streams(args.streams)
    ctx.set_placement_probe(True)      # this process has allocated and freed > 200 GB through torch by now: see mtb_ctx_set_placement_probe (include/mtb.h)
    params = M.default_params(seq_mode=args.seq_mode, syncmer=1, smer_len=5)
    single = rank == 0 and world_size == 1 and not args.partitioned
    do_parity = single and not args.no_parity
    do_legs = single and not args.no_legs and args.seq_mode == 1

    t_setup = time.perf_counter()
    big_world = args.species >= 200
    conserved = big_world and not args.no_conserved
    world = build_world_fast(torch, dev, args.seed, args.species, args.genome_len, args.filler_species, conserved=conserved,
                             n_heldout=args.heldout if ((single and not args.no_legs and args.seq_mode == 1) or args.reads_from == "heldout") else 0) if big_world else \
        build_world(args.seed, args.species, args.genome_len, args.filler_species)
    taxdir = tempfile.mkdtemp(prefix="mtb_tax_")
    world.tax.write(taxdir)
    # N > 1, index replicated, --handover: rank 0 builds the index once and hands it to the other ranks' GPUs (mtb_index_export / mtb_index_import:
    # inter-process handles, device-to-device copies over xGMI -- SURVEY 8(e) row 1 "load once, broadcast"); a rank whose import fails
    # builds its own replica as every rank did before (the result is the same index either way: same seed)
    handover = dist is not None and world_size > 1 and not args.partitioned and args.handover
    index = None; d_values = d_info = None; sealed = False
    handed = dict(mode="local")
    if handover and rank != 0:
        box = [None]
        dist.broadcast_object_list(box, src=0)          # (returns when rank 0 has built, sealed and exported)
        if box[0] is not None:
            try:
                t_i = time.perf_counter()
                index = ctx.import_index(box[0]["share"], taxdir, box[0]["taxid_list"], params)
                torch.cuda.synchronize()
                T, n_real, n_extras, n_filler = box[0]["T"], box[0]["n_real"], box[0]["n_extras"], box[0]["n_filler"]
                sealed = index.state()["sealed"]
                dt_i = time.perf_counter() - t_i
                handed = dict(mode="imported from rank 0", seconds=dt_i, gb_per_s=T * 8 / dt_i / 1e9)
                log(f"[rank {rank}] index imported from rank 0: {T} targets in {dt_i:.2f} s ({T * 8 / dt_i / 1e9:.0f} GB/s)")
            except M.MtbError as e:
                log(f"[rank {rank}] import failed ({e}): building a replica of my own")
                index = None
    if index is None:
        real_v, real_t, n_extras = extract_targets(ctx, M, world, params, torch if big_world else None, dev, hot_min=args.hot_min if conserved else 0, seed=args.seed)
        n_filler = int(args.targets) - len(real_v)
        if n_filler < 0:
            raise SystemExit(f"--targets {int(args.targets)} is the TOTAL: the genomes alone give {len(real_v)} target metamers")
        log(f"[rank {rank}] world: {len(world.genomes)} genomes x {args.genome_len} bp, {len(real_v) - n_extras} genome-derived target metamers + {n_extras} shared-run extras ({time.perf_counter()-t_setup:.1f}s)")
        T_cap = n_filler + len(real_v)
        free, total = torch.cuda.mem_get_info(dev)
        need = T_cap * 12 + args.reads * (args.read_len + 8)
        if need > free * 0.9:
            raise SystemExit(f"index of {T_cap} targets needs {need/2**30:.0f} GiB, only {free/2**30:.0f} GiB free")
        d_values = torch.empty(T_cap, dtype=torch.int64, device=dev)
        d_info = torch.empty(T_cap, dtype=torch.int32, device=dev)
        T = ctx.synth_index(args.seed, n_filler, world.filler_tax_lo, world.filler_tax_hi, real_v, real_t, d_values.data_ptr(), d_info.data_ptr())
        taxid_list = np.concatenate([np.unique(real_t), np.arange(world.filler_tax_lo, world.filler_tax_hi + 1, dtype=np.int32)])
        n_real = len(real_v)
        del real_v, real_t
        index = ctx.index_from_device(d_values.data_ptr(), d_info.data_ptr(), T, taxdir, taxid_list, params)
    if handover and rank == 0:
        box = [None]
        try:
            if not args.no_seal:
                try:
                    index.seal(); sealed = True
                    d_info = None
                    torch.cuda.empty_cache()
                except M.MtbError as e:                   # (an index too small for the packed state is handed over flat)
                    log(f"[rank 0] index not sealed: {e}")
            box[0] = dict(share=index.export(), taxid_list=taxid_list, T=int(T), n_real=int(n_real), n_extras=int(n_extras), n_filler=int(n_filler))
            handed = dict(mode="exported to the other ranks")
        except M.MtbError as e:
            log(f"[rank 0] index not exported ({e}): every rank builds its own replica")
        dist.broadcast_object_list(box, src=0)
    if handover:
        dist.barrier()                                   # the exporter keeps its index open and idle until every importer has its copy
    rseed = args.seed + 17 * (rank + 1)
    d_bases2 = None
    if args.seq_mode == 2:
        d_bases, d_offs, d_bases2 = gen_reads(torch, dev, world.genomes, args.reads, args.read_len, 0.10, 0.005, rseed, paired=True)
    else:
        d_bases, d_offs = gen_reads(torch, dev, world.heldout if args.reads_from == "heldout" else world.genomes, args.reads, args.read_len, 0.10, 0.005, rseed)
    # the other configurations' reads and every parity sample's sub-database: taken from the flat arrays, before anything packs them
    legs = {}
    if do_legs:
        pb, po, pb2 = gen_reads(torch, dev, world.genomes, args.leg_pairs, args.read_len, 0.10, 0.005, rseed + 1, paired=True)
        lb, lo_ = gen_reads(torch, dev, world.genomes, args.leg_long, args.leg_long_len, 0.10, 0.005, rseed + 2)
        legs["paired"] = dict(seq_mode=2, n=args.leg_pairs, read_len=args.read_len, b=pb, o=po, b2=pb2)
        legs["long"] = dict(seq_mode=3, n=args.leg_long, read_len=args.leg_long_len, b=lb, o=lo_, b2=None)
        if big_world:
            bb, bo = gen_reads(torch, dev, world.genomes[:24], args.reads, args.read_len, 0.10, 0.005, rseed + 3)
            legs["best_case"] = dict(seq_mode=1, n=args.reads, read_len=args.read_len, b=bb, o=bo, b2=None)
            if getattr(world, "heldout", None) and args.leg_novel > 0:
                nb_, no_ = gen_reads(torch, dev, world.heldout, args.leg_novel, args.read_len, 0.10, 0.005, rseed + 4)
                legs["novel"] = dict(seq_mode=1, n=args.leg_novel, read_len=args.read_len, b=nb_, o=no_, b2=None)
    sub_main, index_runs = None, None
    if do_parity:
        t_c = time.perf_counter()
        n_s = min(args.reads, args.cpu_reads if args.seq_mode != 3 else max(1, args.cpu_reads * 150 // args.read_len))
        sub_main = sample_closure(ctx, torch, params, d_values, d_info, T, d_bases, d_bases2, args.read_len, n_s, stride=0 if args.no_cpu else args.cpu_stride)
        log(f"[rank 0] sub-database of the headline sample ({n_s} reads, {sub_main['n_kmers']} metamers): {len(sub_main['values'])} targets ({time.perf_counter()-t_c:.1f}s)")
        for name, lg in legs.items():
            if name == "best_case":
                continue
            n_l = min(lg["n"], args.long_parity_reads if lg["seq_mode"] == 3 else args.full_parity_reads)
            lp = M.default_params(seq_mode=lg["seq_mode"], syncmer=1, smer_len=5)
            lg["sub"] = sample_closure(ctx, torch, lp, d_values, d_info, T, lg["b"], lg["b2"], lg["read_len"], n_l)
    if single:
        rl, tl = index.run_histogram()
        index_runs = dict(runs_by_log2_length=rl[:16], targets_by_log2_run_length=tl[:16], quantiles_over_targets=hist_summary(tl),
                          quantiles_over_runs=hist_summary(rl), genome_derived=int(n_real - n_extras), shared_run_extras=int(n_extras), filler=int(n_filler),
                          note="candidate runs (targets sharing one amino-acid part) of the whole timed index; bin b = lengths 2^b .. 2^(b+1)-1")
    if not args.partitioned and not args.no_seal and not sealed:
        # dedicate the index to the fused path: packed 8-byte target words under the amino-acid directory; the info array lent to
        # the library is no longer needed by it and is freed here (64 GB at 16 G targets)
        try:
            index.seal(); sealed = True
            d_info = None
            torch.cuda.empty_cache()
        except M.MtbError as e:
            log(f"[rank {rank}] index not sealed: {e}")
    n_bases_step = args.reads * args.read_len * (2 if args.seq_mode == 2 else 1)
    # (the legs of the other configurations reuse the result arrays: sized for the largest user)
    n_res = max([args.reads] + [lg["n"] for lg in legs.values()])
    d_res = torch.empty(n_res * 24, dtype=torch.uint8, device=dev)
    tc_cap = max([args.reads * (20 + args.read_len // 9) * (2 if args.seq_mode == 2 else 1)] +
                 [lg["n"] * (20 + lg["read_len"] // 9) * (2 if lg["seq_mode"] == 2 else 1) for lg in legs.values()]) + 1024
    d_tt = torch.empty(tc_cap, dtype=torch.int32, device=dev); d_tc = torch.empty(tc_cap, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    log(f"[rank {rank}] setup {time.perf_counter()-t_setup:.1f}s: T={T} ({n_real - n_extras} genome-derived, {n_extras} shared-run extras), reads={args.reads}x{args.read_len}")

    part = None
    if args.partitioned:
        # range r = [bounds[r], bounds[r+1]) cut at amino-acid-part boundaries of the resident array; every rank
        # keeps a device view of its own range only (the full array stays allocated: this mode measures the
        # exchange path, not the capacity gain)
        from metabuli_amd import parallel
        AAM = ~0xFFFFFF
        bounds = np.zeros(world_size, np.uint64)
        for r in range(1, world_size):
            bounds[r] = np.uint64(int(d_values[(T * r) // world_size].item()) & AAM & (2**64 - 1))
        hi = int(bounds[rank + 1]) if rank + 1 < world_size else 2**64 - 1
        part = index.slice(int(bounds[rank]), hi, rank == world_size - 1)
        stages = parallel.GpuStages(ctx, part, params, dev)
        stages.set_reads(d_bases, d_offs, args.reads)
        log(f"[rank {rank}] partitioned: range {rank} holds {part.num_targets} of {T} targets")
        last = {}

    def make_step(p, b, o, b2, n, n_bases):
        return lambda: ctx.classify_batch_device(index, p, b.data_ptr(), o.data_ptr(), b2.data_ptr() if b2 is not None else 0, o.data_ptr() if b2 is not None else 0,
                                                 n, n_bases, d_res.data_ptr(), d_tt.data_ptr(), d_tc.data_ptr(), tc_cap)
    main_step = make_step(params, d_bases, d_offs, d_bases2, args.reads, n_bases_step)

    def step():
        if part is not None:
            last["res"] = parallel.classify_partitioned(stages, bounds, dist)
            return 0
        return main_step()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # The library tries its exact join instantiations on a context's second to fourth batch against an index and keeps the fastest
    # (DESIGN.md section 3): that is set-up, like the index build -- with fewer than 4 warm-up steps the remaining tuning batches run here,
    # untimed, and are reported (`config.tuning_steps`); the W warm-up steps and the K timed steps follow as asked.
    tuning_steps = max(0, 4 - args.warmup) if part is None else 0
    for _ in range(tuning_steps):
        step()
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    sub_batches_timed = ctx.last_sub_batches if part is None else 0      # of the last timed step (the profiled step and the parity sample overwrite the counter)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    st = ctx.last_stats()

    if part is not None:
        res = last["res"][0]
        frac_cls = float((res["is_classified"] != 0).mean())
        log(f"[rank {rank}] partitioned step: classified {frac_cls:.4f}")
        if rank == 0:
            value = args.reads * world_size * args.steps / dt / 1e6
            line = json.dumps(dict(metric="Mreads/s classified (metabuli classify hot path, reads + index resident in HBM)",
                                  value=value, unit="Mreads/s", n_gpus=world_size, steps=args.steps, warmup=args.warmup,
                                  ms_per_step=dt / args.steps * 1e3, higher_is_better=True, scaling="weak", vs_baseline=None,
                                  dtype="u64", data="synthetic",
                                  config=dict(workload=f"{args.reads/1e6:g}M x {args.read_len} bp reads per GPU vs {T/1e9:.2f} G metamers "
                                                       f"range-partitioned over {world_size} GPU(s) (SURVEY 8(e) row 2)",
                                              reads_per_gpu=args.reads, read_len=args.read_len, targets=int(T), seq_mode=args.seq_mode,
                                              classified_fraction=frac_cls,
                                              parallelism=f"index range-partitioned x{world_size}, 2 all-to-all per batch; results to host"),
                                  roofline=None, cpu_baseline=None))
        finish(dist, line if rank == 0 else None)
        return

    if hasattr(M.lib(), "mtb_debug_phase_cycles"):        # profiling build (MTB_LIB=.../libmtb_prof.so): where the join's cycles go
        import ctypes
        pc = (ctypes.c_ulonglong * 24)()
        M.lib().mtb_debug_phase_cycles(ctx.h, pc)
        jt = max(1, sum(pc[16:23]))
        log("[rank 0] k_join_dir phase cycles (thread 0 of every workgroup): " + ", ".join(
            f"{n} {100.0 * pc[16 + i] / jt:.1f} %" for i, n in enumerate(("(rest)", "bisection", "run ends", "wave-scanned runs", "per-lane evaluation + emission",
                                                                         "queries + directory (+ window bounds)", "window staged + barrier"))))
    if hasattr(M.lib(), "mtb_debug_fast_reasons"):        # debugging build (MTB_LIB=.../libmtb_dbg.so): why reads leave the register-resident scorer
        import ctypes
        fr = (ctypes.c_ulonglong * 32)()
        M.lib().mtb_debug_fast_reasons(ctx.h, fr)
        names = ["tail overflow / buckets", "> 24 species rounds (pairs: runs) or > staging", "S1 not sorted", "S2 group of 2", "S3 > 64 paths", "handled"]
        tot = max(1, sum(fr[:6]))
        log("[rank 0] k_score_fast exits (all batches so far): " + ", ".join(f"{n} {fr[i]} ({100.0 * fr[i] / tot:.2f} %)" for i, n in enumerate(names)))
    wl_key = ("diversity" if big_world else "default") + ("" if args.seq_mode == 1 else f"_mode{args.seq_mode}") + ("_heldout" if args.reads_from == "heldout" else "")
    ps, kern, roofline, roofline_all, footprint, query_runs = profiled_step(ctx, M, index, params, step, args.streams, wl_key,
                                                                           (args.reads, args.read_len, int(T), args.seq_mode))

    # sanity of the timed output: fraction of reads classified
    res = np.frombuffer(d_res.cpu().numpy().tobytes(), dtype=M.result_dt)
    frac_cls = float((res["is_classified"] != 0).mean())
    log(f"[rank {rank}] stage ms: extract {st.ms_extract:.1f} sort {st.ms_sort:.1f} join {st.ms_join:.1f} regroup {st.ms_regroup:.1f} "
        f"segsort {st.ms_segsort:.2f} score {st.ms_score:.1f} total {st.ms_total:.1f}; classified {frac_cls:.4f}")
    if frac_cls < 0.5:   # 90 % of the reads come from genomes that are in the index
        raise SystemExit(f"sanity check failed: only {frac_cls:.4f} of the reads were classified")

    # ---- A/B legs of the library's experiment switches (same process, same index, same buffers) ----
    ab = {}

    def ab_legs(tag, fn, n_reads_leg):
        for setting in [x for x in args.ab.split(";") if "=" in x]:
            pairs = [kv.split("=", 1) for kv in setting.split(",") if "=" in kv]          # one leg may set several switches: A=1,B=2
            for k, v in pairs:
                ctx.set_option(k, v)                 # (the library reads its environment once, at mtb_ctx_create: a live context is switched through mtb_ctx_set_option)
            try:
                ms = timed_leg(torch, fn, 1, 3)      # (a forced switch turns the tuner off; the shapes' tuned choices are remembered)
                s2 = ctx.last_stats()
                ab.setdefault(tag, {})[setting] = dict(ms_per_step=ms, mreads_per_s=n_reads_leg / ms / 1e3,
                                                       stage_ms=dict(extract=s2.ms_extract, sort=s2.ms_sort, join=s2.ms_join, score=s2.ms_score, total=s2.ms_total))
                log(f"[rank 0] A/B {tag} {setting}: {ms:.1f} ms per step (join {s2.ms_join:.1f}, score {s2.ms_score:.1f})")
            finally:
                for k, _ in pairs:
                    ctx.set_option(k, os.environ.get(k))
        if args.ab:
            ms = timed_leg(torch, fn, 1, 3)
            s2 = ctx.last_stats()
            ab.setdefault(tag, {})["default"] = dict(ms_per_step=ms, mreads_per_s=n_reads_leg / ms / 1e3,
                                                     stage_ms=dict(extract=s2.ms_extract, sort=s2.ms_sort, join=s2.ms_join, score=s2.ms_score, total=s2.ms_total))
            log(f"[rank 0] A/B {tag} default: {ms:.1f} ms per step (join {s2.ms_join:.1f}, score {s2.ms_score:.1f})")
    if single and args.ab:
        ab_legs("headline", main_step, args.reads)

    # ---- the other configurations, after and outside the timed region: short legs on the SAME sealed index ----
    other = {}
    for name, lg in legs.items():
        lp = M.default_params(seq_mode=lg["seq_mode"], syncmer=1, smer_len=5)
        nb = lg["n"] * lg["read_len"] * (2 if lg["seq_mode"] == 2 else 1)
        lstep = make_step(lp, lg["b"], lg["o"], lg["b2"], lg["n"], nb)
        ms = timed_leg(torch, lstep, 4 if lg["seq_mode"] != 3 else 1, 3 if name != "long" else 2)      # (four untimed steps: the library's join tuning for this batch shape runs on calls 2 - 4)
        ls = ctx.last_stats()
        lres = np.frombuffer(d_res[: lg["n"] * 24].cpu().numpy().tobytes(), dtype=M.result_dt)
        entry = dict(workload=dict(best_case=f"{lg['n']/1e6:g}M x {lg['read_len']} bp single-end reads drawn from 24 of the {len(world.genomes)} genomes (62 x coverage: the round-3 headline's read set) vs the same index",
                                   paired=f"{lg['n']/1e6:g}M x 2 x {lg['read_len']} bp read pairs (BASELINE.json configs[3], per-GPU shape) vs the same index",
                                   long=f"{lg['n']/1e3:g}k x {lg['read_len']} bp long reads (BASELINE.json configs[2] shape) vs the same index",
                                   novel=f"{lg['n']/1e6:g}M x {lg['read_len']} bp single-end reads of {len(getattr(world, 'heldout', []))} HELD-OUT genomes (new species of indexed genera, not in the index: "
                                         f"7.5 % substitutions against the indexed sibling, every conserved segment with its own synonymous redraws) vs the same index -- "
                                         f"no query of a conserved gene finds a target equal to itself, so the join's exact-match shortcut does not apply")[name],
                     seq_mode=lg["seq_mode"], reads=lg["n"], read_len=lg["read_len"], ms_per_step=ms,
                     mreads_per_s=lg["n"] / ms / 1e3, gbp_per_s=nb / ms / 1e6, sub_batches=int(ctx.last_sub_batches),
                     stage_ms=dict(extract=ls.ms_extract, sort=ls.ms_sort, join=ls.ms_join, order=ls.ms_regroup + ls.ms_segsort, score=ls.ms_score, total=ls.ms_total),
                     query_metamers=int(ls.n_kmers), matches=int(ls.n_matches), classified_fraction=float((lres["is_classified"] != 0).mean()),
                     reads_scored_by_generic_kernel=int(ls.n_generic_reads),
                     join_variant=M.JOIN_VARIANTS.get(int(ls.join_variant), str(ls.join_variant)) + (" (tuned)" if ls.join_tuned else ""),
                     reads_deferred=int(ls.n_deferred_reads), reads_scored_by_k_score_many=int(ls.n_many_reads),
                     join_ms_per_G_query_metamers=ls.ms_join / max(1, ls.n_kmers) * 1e9,
                     headline_join_ms_per_G_query_metamers=st.ms_join / max(1, st.n_kmers) * 1e9)
        try:
            lps, lkern, lroof, _, _, lruns = profiled_step(ctx, M, index, lp, lstep, args.streams, wl_key + "_" + name, (lg["n"], lg["read_len"], int(T), lg["seq_mode"]))
            entry["kernel_ms"] = {k: v for k, v in lkern.items() if v["launches"]}
            entry["roofline"] = {k: lroof[k] for k in ("kernel", "achieved", "frac", "traffic", "effective", "frac_design", "avg_launch_ms", "launches", "algorithmic_bytes_per_launch")}
            if lruns is not None:
                entry["query_run_length_quantiles"] = lruns["quantiles"]
                if name == "novel":
                    entry["query_runs"] = lruns
        except M.MtbError as e:
            log(f"[rank 0] leg {name}: no profiled step: {e}")
        if do_parity and "sub" in lg:
            _, lpar = oracle_parity(ctx, M, torch, dev, index, lp, taxdir, lg["b"], lg["b2"], lg["read_len"], lg["sub"], T, label=name)
            entry["parity"] = lpar; entry["mismatches"] = lpar["mismatches"]
            if lpar["mismatches"]:
                if "MTB_NO_SCORE_MANY" not in os.environ:        # diagnosis before giving up: the same sample with the deferred reads on round 4's exact-segment path
                    ctx.set_option("MTB_NO_SCORE_MANY", "1")
                    _, lpar2 = oracle_parity(ctx, M, torch, dev, index, lp, taxdir, lg["b"], lg["b2"], lg["read_len"], lg["sub"], T, label=name + " (MTB_NO_SCORE_MANY=1)")
                    log(f"[rank 0] the same sample without k_score_many: {lpar2['mismatches']} mismatches")
                raise SystemExit(f"parity check of the {name} leg failed: {lpar}")
        log(f"[rank 0] leg {name}: {ms:.1f} ms per step = {entry['mreads_per_s']:.2f} Mreads/s = {entry['gbp_per_s']:.2f} Gbp/s")
        other[name] = entry
        if args.ab and name in ("best_case", "novel", "paired"):
            ab_legs(name, lstep, lg["n"])
    best_case = other.pop("best_case", None)

    cpu, parity = None, None
    if do_parity:
        cpu, parity = oracle_parity(ctx, M, torch, dev, index, params, taxdir, d_bases, d_bases2, args.read_len, sub_main, T, time_cpu=not args.no_cpu, label="headline")
        if parity["mismatches"]:
            raise SystemExit(f"parity check against the timed index failed: {parity}")

    # which library was timed, on which device, per rank (SCALE runs are audited with this)
    ident = dict(rank=rank, index=handed, device=f"cuda:{local_rank}" if device is None else str(dev), library=os.path.realpath(M.LIB_PATH), version=M.lib().mtb_version().decode(),
                 device_name=(torch.cuda.get_device_name(local_rank) if device is None else "emulated"))
    log(f"[rank {rank}] library {ident['library']} ({ident['version']}) on {ident['device']} ({ident['device_name']})")
    ranks = [ident]
    if dist is not None:
        ranks = [None] * world_size
        dist.all_gather_object(ranks, ident)
    peak_measured = copy_peak_gbs(torch, dev) if (rank == 0 and device is None) else None
    if rank == 0:
        roofline["peak_measured"] = peak_measured
        roofline["peak_measured_note"] = "device-to-device copy of 4 GiB measured in this run (read + write bytes / time): the practical HBM ceiling next to the data-sheet `peak`"
        if peak_measured:
            roofline["frac_of_measured_peak"] = roofline["achieved"] / peak_measured if roofline["achieved"] is not None else None
            if roofline.get("effective") is not None:
                roofline["effective_of_measured_peak"] = roofline["effective"] * PEAK_GBS / peak_measured
        total_reads = args.reads * world_size * args.steps
        value = total_reads / dt / 1e6
        cfg_name = {1: 'BASELINE.json configs[1]', 2: 'BASELINE.json configs[3] shape: paired-end, index replicated, reads sharded', 3: 'BASELINE.json configs[2] shape: long reads'}[args.seq_mode]
        out = dict(metric="Mreads/s classified (metabuli classify hot path, reads + index resident in HBM)",
                   value=value, unit="Mreads/s", n_gpus=world_size, steps=args.steps, warmup=args.warmup,
                   ms_per_step=dt / args.steps * 1e3, higher_is_better=True, scaling="weak", vs_baseline=None,
                   dtype="u64", data="synthetic",
                   config=dict(workload=f"{args.reads/1e6:g}M x {'2 x ' if args.seq_mode == 2 else ''}{args.read_len} bp synthetic "
                                        f"{ {1: 'single-end', 2: 'paired-end', 3: 'long'}[args.seq_mode] } reads per GPU from {len(world.genomes)} "
                                        f"{'held-out ' if args.reads_from == 'heldout' else ''}genomes vs synthetic GTDB-scale index of {T/1e9:.2f} G metamers "
                                        f"({T*12/2**30:.0f} GiB flat, replicated per GPU{', heavy-tailed candidate runs' if conserved else ''}), syncmer s=5, kmer_format 2 ({cfg_name})",
                               reads_per_gpu=args.reads, read_len=args.read_len, targets=int(T), seq_mode=args.seq_mode,
                               gbp_per_s=value * args.read_len * (2 if args.seq_mode == 2 else 1) / 1e3, query_metamers=int(st.n_kmers), matches=int(st.n_matches),
                               classified_fraction=frac_cls, parallelism=f"reads sharded x{world_size}, index replicated", streams_per_gpu=args.streams,
                               sub_batches_per_step=sub_batches_timed, index_sealed=sealed, tuning_steps=tuning_steps,
                               index_handover=(f"{sum(1 for x in ranks if x['index']['mode'].startswith('imported'))} of {world_size - 1} ranks imported rank 0's index" if world_size > 1 else None),
                               join_variant=M.JOIN_VARIANTS.get(int(st.join_variant), str(st.join_variant)) + (" (tuned)" if st.join_tuned else ""),
                               join_tune_ms=dict(zip(("q1w6", "q2w5", "window"), [round(float(x), 2) for x in st.join_tune_ms])),
                               join_tiles=dict(tiles=int(st.join_tiles), windowed=int(st.join_tiles_windowed), outside=int(st.join_tiles_outside)),
                               index_bytes=int(T * (8 if sealed else 12) + 4 * (21 ** index.state()["dir_depth"] + 1)), species=args.species, genome_len=args.genome_len,
                               conserved_segments=conserved, reads_scored_by_generic_kernel=int(ps.n_generic_reads), reads_on_ordinal_slots=int(ps.n_slot_reads)),
                   stage_ms=dict(extract=st.ms_extract, sort=st.ms_sort, join=st.ms_join, regroup=st.ms_regroup,
                                 segsort=st.ms_segsort, score=st.ms_score, total=st.ms_total),
                   kernel_ms=kern, roofline=roofline, roofline_all=roofline_all, join_footprint=footprint,
                   run_lengths=dict(index=index_runs, queries=query_runs),
                   best_case=best_case, other_configs=other,
                   cpu_baseline=cpu, parity_sample=parity, parity_full_index=parity,
                   library=dict(path=ident["library"], version=ident["version"]), ranks=ranks, ab=ab or None,
                   deferred_reads=dict(deferred_by_the_slot_scorers=int(ps.n_deferred_reads), scored_by_k_score_many=int(ps.n_many_reads),
                                       their_matches=int(ps.n_many_matches), survivors_of_the_dead_species_drop=int(ps.n_many_kept),
                                       note="reads whose tails overflow (conserved genes: hundreds of matches over hundreds of species): k_score_many takes them from "
                                            "slots + overflow entries and drops the species without a (species, frame) group of two before anything is ordered"))
        finish(dist, emit_lines(out))
    else:
        finish(dist, None)


if __name__ == "__main__":
    main()
