"""Brute-force evaluators of the hot path's *formulas* (SURVEY.md 8a), in plain Python / numpy.

TEST INFRASTRUCTURE ONLY.  A third, independent formulation next to oracle/oracle.cpp (which follows the reference's
control flow: streaming scanners, streaming merge, Taxonomer loops) and metabuli_amd/csrc/mtb_core.h (the per-lane
kernel arithmetic): nothing here calls, includes or was derived from either.  The constant tables are read from the
fixtures that are pinned to the reference itself (tests/golden/ref_codon_tables.txt = output of the reference's own
GeneticCode.h; tests/golden/ref_hamming_tables.json = the numeric literals of KmerMatcher.h:66-158).

  extract_spec   a2-a4  every window of 8 valid codons of the six reading frames, closed-syncmer filter by exhaustive
                        minimum over the s-mers of the window (KmerExtractor.cpp:342-373, KmerScanner.h:82-117,
                        SyncmerScanner.h:36-101)
  decode_diffidx a8     the delta stream, vectorised (KmerMatcher.h:282-297)
  join_spec      a10-11 Matches(q) = {t < T-1 : AA(t) = AA(q), ham(q,t) <= min(2 min ham, 7)} by searchsorted + all-pairs
                        Hamming (KmerMatcher.cpp:123-481, 1117-1146)
  score_spec     a13-18 per read: every chain of consecutive matches is enumerated explicitly (no dynamic programming),
                        the best chain per end match chosen by exhaustive comparison; then the combination, the species
                        decision, the redundancy filter and the sub-species descent (Taxonomer.cpp:130-699)
"""
from __future__ import annotations

import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
AAMASK = np.uint64(0xFFFFFFFFFF000000)


# --------------------------------------------------------------------------------------------------
# tables pinned to the reference
# --------------------------------------------------------------------------------------------------
def ref_tables():
    rows = [l for l in open(os.path.join(HERE, "golden", "ref_codon_tables.txt")) if not l.startswith("#")]
    aa = np.array(rows[0].split(), dtype=np.int64).reshape(8, 8, 8)
    num = np.array(rows[1].split(), dtype=np.int64).reshape(8, 8, 8)
    fwd = np.array(rows[2].split(), dtype=np.int64)          # nuc2int(atcg[c])
    rev = np.array(rows[3].split(), dtype=np.int64)          # nuc2int(iRCT[atcg[c]])
    ham = json.load(open(os.path.join(HERE, "golden", "ref_hamming_tables.json")))
    lookup = np.array(ham["hammingLookup"], dtype=np.int64).reshape(8, 8)
    luts = [np.array(ham[f"HAMMING_LUT{i}"], dtype=np.int64) for i in range(8)]
    return dict(aa=aa, num=num, fwd=fwd, rev=rev, lookup=lookup, luts=luts)


def used_len(L):
    """LocalUtil.h:51-59"""
    return L - 2 if L % 3 == 2 else (L - 4 if L % 3 == 1 else L - 3)


# --------------------------------------------------------------------------------------------------
# a2-a4: extraction (kmer_format 2)
# --------------------------------------------------------------------------------------------------
def extract_read_spec(T, seq: bytes, seq_id: int, syncmer: int, smer_len: int, offset: int = 0):
    """All (value, qinfo) of one read, kmer_format 2.  A window is 8 consecutive codons of one frame, all valid."""
    L = len(seq)
    out = []
    used = used_len(L)
    if (used // 3 - 8 + 1) * 6 < 1:
        return out
    f = T["fwd"][np.frombuffer(seq, dtype=np.uint8)]
    r = T["rev"][np.frombuffer(seq, dtype=np.uint8)]
    for frame in range(6):
        fwd = frame < 3
        begin = frame if fwd else ((L % 3) - (frame % 3)) % 3
        end = begin + used - 1                      # inclusive window of the scanner
        n_cod = used // 3
        cod_aa, cod_id = [], []
        for j in range(n_cod):
            if fwd:
                a, b, c = f[begin + 3 * j], f[begin + 3 * j + 1], f[begin + 3 * j + 2]
            else:                                   # complement strand, read from the right end
                a, b, c = r[end - 3 * j], r[end - 3 * j - 1], r[end - 3 * j - 2]
            if max(a, b, c) > 3:
                cod_aa.append(-1); cod_id.append(-1)
            else:
                cod_aa.append(int(T["aa"][a, b, c])); cod_id.append(int(T["num"][a, b, c]))
        for p in range(n_cod - 8 + 1):
            aas, ids = cod_aa[p:p + 8], cod_id[p:p + 8]
            if min(aas) < 0:
                continue
            if syncmer:
                smers = [tuple(aas[k:k + smer_len]) for k in range(8 - smer_len + 1)]
                arg = smers.index(min(smers))       # leftmost minimum
                if arg != 0 and arg != 8 - smer_len:
                    continue
            aa_part = 0
            dna = 0
            for a_, i_ in zip(aas, ids):
                aa_part = (aa_part << 5) | a_
                dna = (dna << 3) | i_
            value = (aa_part << 24) | dna
            pos = begin + 3 * p if fwd else end - 3 * (p + 8) + 1
            out.append((value, (pos + offset) | (seq_id << 32) | (frame << 61)))
    return out


def extract_spec(T, bases, offs, bases2=None, offs2=None, syncmer=1, smer_len=5, reads=None):
    """(value, qinfo) arrays of a batch (seq_mode 1/2/3 geometry: mates share the sequenceID, mate 2 positions are shifted
    by used(L1) + 3, a pair with a too-short mate is skipped, KmerExtractor.cpp:322-329, 443-453)."""
    vals, qis = [], []
    n = len(offs) - 1
    for i in (range(n) if reads is None else reads):
        s1 = bytes(bases[int(offs[i]):int(offs[i + 1])])
        short1 = (used_len(len(s1)) // 3 - 8 + 1) * 6 < 1
        if bases2 is not None:
            s2 = bytes(bases2[int(offs2[i]):int(offs2[i + 1])])
            if short1 or (used_len(len(s2)) // 3 - 8 + 1) * 6 < 1:
                continue
        elif short1:
            continue
        recs = extract_read_spec(T, s1, i + 1, syncmer, smer_len)
        if bases2 is not None:
            recs += extract_read_spec(T, s2, i + 1, syncmer, smer_len, offset=used_len(len(s1)) + 3)
        for v, q in recs:
            vals.append(v); qis.append(q)
    return np.array(vals, dtype=np.uint64), np.array(qis, dtype=np.uint64)


# --------------------------------------------------------------------------------------------------
# a8: diffIdx
# --------------------------------------------------------------------------------------------------
def decode_diffidx(d16: np.ndarray) -> np.ndarray:
    """u16 stream -> target values: 15-bit groups, big-endian, the last group of a metamer has bit 15 set; every metamer
    is stored as the difference to its predecessor (the first one to 0)."""
    d16 = np.asarray(d16, dtype=np.uint16)
    last = (d16 & 0x8000) != 0
    ends = np.flatnonzero(last)
    if len(ends) == 0:
        return np.zeros(0, np.uint64)
    starts = np.concatenate([[0], ends[:-1] + 1])
    frag = (d16 & 0x7FFF).astype(np.uint64)
    deltas = np.zeros(len(ends), np.uint64)
    width = ends - starts + 1
    for k in range(int(width.max())):               # group k counted from the END of each metamer
        has = width > k
        deltas[has] |= frag[ends[has] - k] << np.uint64(15 * k)
    return np.cumsum(deltas, dtype=np.uint64)


# --------------------------------------------------------------------------------------------------
# a10-a11: join
# --------------------------------------------------------------------------------------------------
def _codons(x):
    return [(int(x) >> (3 * i)) & 7 for i in range(8)]


def join_spec(T, values, info, species_of, q_values, q_infos, kmer_format=2, info_mask=0xFFFFFFFF):
    """-> list of match tuples (qinfo, target_id, species_id, dna, right_end_hamming, hamming), unsorted."""
    values = np.asarray(values, dtype=np.uint64)
    cand_values = values[:-1] if len(values) else values          # the last entry of the index is never a candidate
    cand_aa = cand_values & AAMASK
    lookup, luts = T["lookup"], T["luts"]
    out = []
    for qv, qi in zip(q_values, q_infos):
        if (int(qi) >> 32) & 0x1FFFFFFF == 0:
            continue
        aa = np.uint64(qv) & AAMASK
        lo = int(np.searchsorted(cand_aa, aa, side="left")); hi = int(np.searchsorted(cand_aa, aa, side="right"))
        if lo == hi:
            continue
        qc = _codons(int(qv) & 0xFFFFFF)
        hams = []
        for t in range(lo, hi):
            tc = _codons(int(values[t]) & 0xFFFFFF)
            hams.append(sum(int(lookup[a, b]) for a, b in zip(qc, tc)))
        thr = min(2 * min(hams), 7)
        frame = int(qi) >> 61
        plain = not ((frame < 3) ^ (kmer_format == 2))             # getHammings, else getHammings_reverse
        for t, h in zip(range(lo, hi), hams):
            if h > thr:
                continue
            tc = _codons(int(values[t]) & 0xFFFFFF)
            h16 = 0
            for i in range(8):
                lut = luts[i] if plain else luts[7 - i]
                h16 |= int(lut[(qc[i] << 3) | tc[i]])
            tid = int(info[t]) & info_mask
            tid = tid - (1 << 32) if tid >= (1 << 31) else tid
            out.append((int(qi), tid, species_of(tid), int(values[t]) & 0xFFFFFF, h16, h))
    return out


def sort_matches_spec(matches):
    """compareMatches (KmerMatcher.cpp:1149-1166): (sequenceID, species, frame, position, hamming, dna)"""
    return sorted(matches, key=lambda m: ((m[0] >> 32) & 0x1FFFFFFF, m[2], m[0] >> 61, m[0] & 0xFFFFFFFF, m[5], m[3]))


# --------------------------------------------------------------------------------------------------
# a13-a18: scoring of one read by explicit enumeration
# --------------------------------------------------------------------------------------------------
def _cscore(h):
    return 3.0 if h == 0 else 2.0 - 0.5 * h


class TooManyChains(Exception):
    pass


class ReadScorer:
    """tax: synth.Taxonomy-like object with .parent/.rank dicts and .lca(a, b)."""

    def __init__(self, tax, syncmer, smer_len, seq_mode, kmer_format=2, min_cons_cnt=4, min_cons_cnt_euk=9, min_score=0.0, min_sp_score=0.0,
                 tie_ratio=0.95, accession_level=0, eukaryota=0, max_chains=200000):
        self.tax = tax
        self.max_shift = 8 - smer_len if syncmer else 1
        self.dna_shift = (8 - smer_len) * 3 if syncmer else 3
        self.denominator = 100 if seq_mode in (1, 2) else 1000
        self.kf = kmer_format
        self.mcc, self.mcc_euk = min_cons_cnt, min_cons_cnt_euk
        self.min_score, self.min_sp_score, self.tie_ratio = np.float32(min_score), np.float32(min_sp_score), np.float32(tie_ratio)
        self.acc = accession_level
        self.euk = eukaryota
        self.max_chains = max_chains

    # -- taxonomy helpers --------------------------------------------------------------------------
    def _is_under(self, anc, t):
        if anc == 0 or t == 0 or t not in self.tax.parent:
            return anc == t
        while True:
            if t == anc:
                return True
            p = self.tax.parent[t]
            if p == t:
                return False
            t = p

    def _lca_list(self, ids):
        cur = None
        for t in ids:
            if t not in self.tax.parent:
                continue
            cur = t if cur is None else self.tax.lca(cur, t)
        return 0 if cur is None else cur

    def _lca2(self, a, b):
        if a not in self.tax.parent:
            return b
        if b not in self.tax.parent:
            return a
        return self.tax.lca(a, b)

    # -- a15: all chains of one (species, frame) block ----------------------------------------------
    def _consecutive(self, cur, nxt, shift, fwd):
        a, b = (cur[3], nxt[3]) if fwd else (nxt[3], cur[3])
        keep = (1 << (24 - 3 * shift)) - 1
        if self.kf == 2:
            return (a & keep) == (b >> (3 * shift))
        return (a >> (3 * shift)) == (b & keep)

    def block_paths(self, block, species):
        """block: matches of one species and one frame, sorted by position.  Returns the emitted paths in emission order as
        dicts(start, end, score, ham, depth, start_match, end_match)."""
        if len(block) < 2:                                      # Taxonomer.cpp:342
            return []
        min_depth = self.mcc_euk if (self.euk and self._is_under(self.euk, species)) else self.mcc
        fwd = (block[0][0] >> 61) < 3
        pos = [m[0] & 0xFFFFFFFF for m in block]
        groups = []
        for i, p in enumerate(pos):
            if groups and pos[groups[-1][0]] == p:
                groups[-1].append(i)
            else:
                groups.append([i])
        if len(groups) < 2:                                     # paths are only pushed while walking to a next position
            return []
        # edges between adjacent position groups
        preds = {i: [] for i in range(len(block))}
        connected = set()
        for g in range(len(groups) - 1):
            shift = (pos[groups[g + 1][0]] - pos[groups[g][0]]) // 3
            if not (0 < shift <= self.max_shift):
                continue
            for nx in groups[g + 1]:
                for cu in groups[g]:
                    if self._consecutive(block[cu], block[nx], shift, fwd):
                        preds[nx].append((cu, shift))
                        connected.add(cu)
        budget = [self.max_chains]

        def chains_to(i):
            """every chain ending at match i, as (score, index tuple from the end backwards, ham, depth, first index)"""
            m = block[i]
            own = (sum(_cscore((m[4] >> (2 * c)) & 3) for c in range(8)), (i,), m[5], 1, i)
            res = [own] if not preds[i] else []
            for cu, shift in preds[i]:
                inc = sum(_cscore((m[4] >> (2 * c)) & 3) for c in range(shift))
                hinc = sum((m[4] >> (2 * c)) & 3 for c in range(shift))
                for (s, idx, h, d, first) in chains_to(cu):
                    budget[0] -= 1
                    if budget[0] < 0:
                        raise TooManyChains()
                    res.append((s + inc, (i,) + idx, h + hinc, d + shift, first))
            return res

        emitted = []
        order = []                                              # emission order: group by group, a group when it is left
        for g in range(len(groups)):
            for i in groups[g]:
                if g + 1 < len(groups):
                    if i not in connected:
                        order.append(i)
                else:
                    order.append(i)
        for i in order:
            cands = chains_to(i)
            # the chain the reference's forward pass ends up with: highest score; among equals the one whose
            # predecessors come first in list order, compared from the end match backwards
            best = max(cands, key=lambda c: (c[0], tuple(-x for x in c[1])))
            if best[3] >= min_depth:
                emitted.append(dict(start=block[best[4]][0] & 0xFFFFFFFF, end=(block[i][0] & 0xFFFFFFFF) + 23, score=np.float32(best[0]),
                                    ham=best[2], depth=best[3], start_match=block[best[4]], end_match=block[i]))
        return emitted

    # -- a16 ------------------------------------------------------------------------------------------
    @staticmethod
    def combine(paths, read_len):
        paths = sorted(paths, key=lambda p: (-float(p["score"]), p["ham"], -p["start"]))       # stable, like a small std::sort
        acc = []
        total = np.float32(0)
        for p in paths:
            p = dict(p)
            drop = False
            for c in acc:
                if p["end"] < c["start"] or c["end"] < p["start"]:
                    continue
                ov = min(p["end"], c["end"]) - max(p["start"], c["start"]) + 1
                if ov == p["end"] - p["start"] + 1 or ov >= 24:
                    drop = True
                    break
                k = ov // 3
                if p["start"] < c["start"]:
                    reh = p["end_match"][4]
                    p["end"] = c["start"] - 1
                    p["ham"] = max(0, p["ham"] - sum((reh >> (2 * i)) & 3 for i in range(k)))
                    p["score"] = np.float32(p["score"] - np.float32(sum(_cscore((reh >> (2 * i)) & 3) for i in range(k))) - np.float32(ov % 3))
                else:
                    reh = p["start_match"][4]
                    p["start"] = c["end"] + 1
                    p["ham"] = max(0, p["ham"] - sum((reh >> (14 - 2 * i)) & 3 for i in range(k)))
                    p["score"] = np.float32(p["score"] - np.float32(sum(_cscore((reh >> (14 - 2 * i)) & 3) for i in range(k))) - np.float32(ov % 3))
            if not drop:
                acc.append(p)
                total = np.float32(total + p["score"])
        return np.float32(total / np.float32(read_len))

    # -- a17-a18 ---------------------------------------------------------------------------------------
    def _lower_rank(self, taxcnt, species, read_len):
        thr = (read_len - 1) // self.denominator
        clade, children = {}, {}
        for t, c in taxcnt.items():
            node = t
            clade[node] = clade.get(node, 0) + c
            while node != species:
                par = self.tax.parent[node]
                children.setdefault(par, [])
                if node not in children[par]:
                    children[par].append(node)
                clade[par] = clade.get(par, 0) + c
                node = par
        if self.acc == 2:
            for t in list(clade):
                if self.tax.rank.get(t, "") in ("", "accession"):
                    par = self.tax.parent[t]
                    if t in children.get(par, []):
                        children[par].remove(t)
        root = species
        while True:
            ch = children.get(root, [])
            if not ch:
                return root
            mx, best = thr, []
            for c in ch:
                if clade[c] > mx:
                    best, mx = [c], clade[c]
                elif clade[c] == mx:
                    best.append(c)
            if len(best) != 1:
                return root
            root = best[0]

    def score_read(self, matches, qlen, qlen2):
        """matches of one read in compareMatches order -> (classification, score f32, is_classified, taxcnt dict)"""
        read_len = qlen + qlen2
        sp2score = []
        best_sp, best_range, meaningful = np.float32(0), None, 0
        i = 0
        while i < len(matches):
            sp = matches[i][2]
            s = i
            paths = []
            while i < len(matches) and matches[i][2] == sp:
                fr = matches[i][0] >> 61
                b = i
                while i < len(matches) and matches[i][2] == sp and (matches[i][0] >> 61) == fr:
                    i += 1
                paths += self.block_paths(matches[b:i], sp)
            if not paths:
                continue
            sc = min(self.combine(paths, read_len), np.float32(1.0))
            if sc < self.min_score:
                continue
            sp2score.append((sp, sc))
            if sc > 0:
                meaningful += 1
            if sc > best_sp:
                best_sp, best_range = sc, (s, i)
        if meaningful == 0:
            return 0, np.float32(0), 0, {}
        tied = [(sp, sc) for sp, sc in sp2score if sc >= np.float32(best_sp * self.tie_ratio)]
        score = np.float32(0)
        for _, sc in tied:
            score = np.float32(score + sc)
        if len(tied) > 1:
            score = np.float32(score / np.float32(len(tied)))
        if score == 0 or score < self.min_score:
            return 0, score, 0, {}
        if len(tied) > 1:
            return self._lca_list([sp for sp, _ in tied]), score, 1, {}
        species = tied[0][0]
        # filterRedundantMatches: per position bucket the LCA of the minimum-hamming matches
        buckets = {}
        for m in matches[best_range[0]:best_range[1]]:
            q = (m[0] & 0xFFFFFFFF) // self.dna_shift
            if q not in buckets or m[5] < buckets[q][1]:
                buckets[q] = [m[1], m[5]]
            elif m[5] == buckets[q][1]:
                buckets[q][0] = self._lca2(buckets[q][0], m[1])
        taxcnt = {}
        for t, _ in buckets.values():
            taxcnt[t] = taxcnt.get(t, 0) + 1
        if score < self.min_sp_score:
            return self.tax.parent[species], score, 1, taxcnt
        return self._lower_rank(taxcnt, species, read_len), score, 1, taxcnt
