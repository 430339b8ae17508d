"""Pins the oracle's restated tables against the reference: (a) the committed
output of the reference's own GeneticCode.h compiled verbatim
(tests/golden/ref_codon_tables.txt, regenerated live from oracle/_ref when the
reference tree is present), (b) the numeric literals of KmerMatcher.h:66-158."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src/commons"


def _load_ref_codon(path):
    lines = [l for l in open(path).read().split("\n") if l and not l.startswith("#")]
    aa = np.array(lines[0].split(), dtype=np.int64).reshape(8, 8, 8)
    num = np.array(lines[1].split(), dtype=np.int64).reshape(8, 8, 8)
    fwd = np.array(lines[2].split(), dtype=np.int64)
    rev = np.array(lines[3].split(), dtype=np.int64)
    return aa, num, fwd, rev


def _check_codon(orc, path):
    raa, rnum, rfwd, rrev = _load_ref_codon(path)
    aa, num = orc.codon_tables()
    idx = [0, 1, 2, 3, 7]                 # the reference leaves indices 4-6 uninitialised
    for a in idx:
        for b in idx:
            for c in idx:
                assert raa[a, b, c] == aa[a, b, c], (a, b, c)
                if raa[a, b, c] >= 0:
                    assert rnum[a, b, c] == num[a, b, c], (a, b, c)
    f, r = orc.base_codes()
    assert (rfwd == f).all() and (rrev == r).all()


def test_codon_tables_vs_committed_reference_dump(orc):
    _check_codon(orc, os.path.join(HERE, "golden", "ref_codon_tables.txt"))


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "GeneticCode.h")), reason="reference tree not present")
def test_codon_tables_vs_live_reference_build(orc):
    import subprocess
    root = os.path.dirname(HERE)
    subprocess.check_call(["make", "-C", os.path.join(root, "oracle"), "ref"], stdout=subprocess.DEVNULL)
    _check_codon(orc, os.path.join(root, "oracle", "_ref", "ref_codon_tables.txt"))


def test_hamming_tables_vs_reference_literals(orc):
    ref = json.load(open(os.path.join(HERE, "golden", "ref_hamming_tables.json")))
    lk, lut = orc.hamming_tables()
    assert (lk.reshape(-1) == np.array(ref["hammingLookup"])).all()
    for k in range(8):
        assert (lut[k] == np.array(ref["HAMMING_LUT%d" % k])).all(), k


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "KmerMatcher.h")), reason="reference tree not present")
def test_committed_hamming_fixture_is_current():
    import re
    src = open(os.path.join(REF, "KmerMatcher.h")).read()
    ref = json.load(open(os.path.join(HERE, "golden", "ref_hamming_tables.json")))
    m = re.search(r"HAMMING_LUT7\[64\]\s*=\s*\{(.*?)\};", src, re.S)
    body = re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S)
    assert [int(x) for x in re.findall(r"\d+", body)] == ref["HAMMING_LUT7"]


def test_kernel_tables_equal_oracle(orc, emu):
    """The tables libmtb uploads to the GPU (mtb_build_tables) against the pinned oracle."""
    base, codon, hamrow = emu.tables()
    f, r = orc.base_codes()
    assert (base == f).all()
    aa, num = orc.codon_tables()
    for a in range(4):
        for b in range(4):
            for c in range(4):
                v = int(codon[a * 16 + b * 4 + c])
                assert (v & 31) == aa[a, b, c] and (v >> 5) == num[a, b, c]
    lk, _ = orc.hamming_tables()
    for a in range(8):
        for b in range(8):
            assert (int(hamrow[a]) >> (4 * b)) & 15 == lk[a, b]


def test_hamming_vectors_random(orc, emu):
    """getHammingDistanceSum / getHammings(_reverse) on random codon strings through
    the kernel arithmetic (via the join of a one-entry run) == oracle."""
    rng = np.random.default_rng(0)
    lk, lut = orc.hamming_tables()
    for _ in range(2000):
        a = int(rng.integers(0, 1 << 24)); b = int(rng.integers(0, 1 << 24))
        s = sum(int(lk[(a >> (3 * i)) & 7, (b >> (3 * i)) & 7]) for i in range(8))
        assert orc.lib.orc_hamming_sum(a, b) == s
        h = 0
        for i in range(8):
            h |= int(lut[i][(((a >> (3 * i)) & 7) << 3) | ((b >> (3 * i)) & 7)])
        assert orc.lib.orc_hammings(a, b) == h
