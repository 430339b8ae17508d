"""Binary `taxonomyDB`: the reader libmtb uses at index-open time (host_db.h::load_taxonomy_db, built for
the host in tests/emu) against the dump-file loader on the same taxonomy, through fixtures written by tests/taxdb_writer.py
(TaxonomyWrapper::serialize restated).  Layout of the MMseqs2 parts: restated from published sources, unpinned."""
import ctypes as C
import os

import numpy as np
import pytest

import taxdb_writer as tw
from helpers import _ptr


def _world(seed=5):
    from metabuli_amd import synth
    w = synth.make_world(seed=seed, n_genera=3, species_per_genus=2, strains_per_species=2, genome_len=2000)
    return w


def _orig_id(t):
    return 1 if t == 1 else 1000 + 7 * t          # original ids far from the dense internal ones; the root stays 1


def _write_dmp(d, w, merged=()):
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "nodes.dmp"), "w") as f:
        for t in sorted(w.tax.parent):
            f.write(f"{_orig_id(t)}\t|\t{_orig_id(w.tax.parent[t])}\t|\t{w.tax.rank[t]}\t|\t\t|\n")
    with open(os.path.join(d, "names.dmp"), "w") as f:
        for t in sorted(w.tax.parent):
            f.write(f"{_orig_id(t)}\t|\t{w.tax.name[t]}\t|\t\t|\tscientific name\t|\n")
    with open(os.path.join(d, "merged.dmp"), "w") as f:
        for a, b in merged:
            f.write(f"{a}\t|\t{b}\t|\n")


def _lines(w):
    return [(_orig_id(t), _orig_id(w.tax.parent[t]), w.tax.rank[t]) for t in sorted(w.tax.parent)], {_orig_id(t): w.tax.name[t] for t in w.tax.parent}


def _load_db(emu, path, ids):
    cap = 1 << 20
    a = [np.zeros(cap, np.int32) for _ in range(3)]
    under = np.zeros(cap, np.uint8); spp = np.zeros(cap, np.int32); t2s = np.zeros(cap, np.int32); acc = np.zeros(cap, np.uint8); orig = np.zeros(cap, np.int32)
    mx = C.c_int32(); euk = C.c_int32()
    err = C.create_string_buffer(512)
    ids = np.ascontiguousarray(ids, np.int32)
    rc = emu.lib.emu_load_taxonomy_db(path.encode(), _ptr(ids), C.c_size_t(len(ids)), C.c_int32(cap), C.byref(mx), _ptr(a[0]), _ptr(a[1]), _ptr(a[2]),
                                      _ptr(under), _ptr(spp), _ptr(t2s), _ptr(acc), _ptr(orig), C.byref(euk), err, C.c_size_t(512))
    if rc:
        return rc, err.value.decode()
    n = mx.value + 1
    return 0, dict(canon=a[0][:n].copy(), parent=a[1][:n].copy(), depth=a[2][:n].copy(), under=under[:n].copy(), spp=spp[:n].copy(), t2s=t2s[:n].copy(),
                   acc=acc[:n].copy(), orig=orig[:n].copy(), euk=euk.value)


@pytest.mark.parametrize("k_extra", [0, 1])
def test_taxonomy_db_equals_dump_files(emu, tmp_path, k_extra):
    w = _world()
    merged = [(900001, _orig_id(5)), (900002, _orig_id(9))]
    d = str(tmp_path / "tax")
    _write_dmp(d, w, merged)
    lines, names = _lines(w)
    path = str(tmp_path / "taxonomyDB")
    o2i = tw.write_taxonomy_db(path, lines, names, merged, use_internal=True, k_extra=k_extra)
    strains = [t for t, r in w.tax.rank.items() if r == "no rank" and t > 3]
    ids_orig = np.array([_orig_id(t) for t in strains], np.int32)
    (canon, parent, depth, under, spp, acc), t2s = emu.load_taxonomy(d, ids_orig)
    rc, db = _load_db(emu, path, np.array([o2i[int(o)] for o in ids_orig], np.int32))
    assert rc == 0, db
    assert db["orig"][0] == 0 and db["euk"] == o2i[_orig_id(3)]
    assert len(db["canon"]) == len(o2i) + 1                      # internal ids are dense: 1..maxTaxID
    for o, i in o2i.items():
        assert db["orig"][i] == o
        co = canon[o]
        if co < 0:
            assert db["canon"][i] < 0
            continue
        ci = db["canon"][i]
        assert db["orig"][ci] == co                                # merged ids resolve to the same node
        if co != o:
            continue
        assert db["orig"][db["parent"][i]] == parent[o] and db["depth"][i] == depth[o] and db["under"][i] == under[o] and db["acc"][i] == acc[o]
        assert db["orig"][db["spp"][i]] == spp[o] if spp[o] else db["spp"][i] == 0
        assert (db["orig"][db["t2s"][i]] if db["t2s"][i] else 0) == t2s[o]
    assert db["canon"][o2i[900001]] == o2i[_orig_id(5)]


def test_taxonomy_db_without_internal_ids_and_bad_version(emu, tmp_path):
    w = _world(7)
    lines, names = _lines(w)
    path = str(tmp_path / "plain")
    tw.write_taxonomy_db(path, lines, names, use_internal=False)          # ends in the unused id-map bytes, as the reference's files do
    rc, db = _load_db(emu, path, np.zeros(0, np.int32))
    assert rc == 0, db
    tight = str(tmp_path / "plain_tight")
    tw.write_taxonomy_db(tight, lines, names, use_internal=False, slack=False)
    rc, db2 = _load_db(emu, tight, np.zeros(0, np.int32))
    assert rc == 0 and (db2["canon"] == db["canon"]).all() and (db2["parent"] == db["parent"]).all()
    present = np.flatnonzero(db["canon"] >= 0)
    assert set(present.tolist()) == {o for o, _, _ in lines}
    assert (db["orig"] == np.arange(len(db["orig"]))).all()      # getOriginalTaxID is the identity without internal ids
    bad = str(tmp_path / "old")
    tw.write_taxonomy_db(bad, lines, names, version=1)
    rc, msg = _load_db(emu, bad, np.zeros(0, np.int32))
    assert rc == 1 and "version" in msg
    trunc = str(tmp_path / "trunc")
    open(trunc, "wb").write(open(path, "rb").read()[:-7])
    rc, msg = _load_db(emu, trunc, np.zeros(0, np.int32))
    assert rc == 1


@pytest.mark.parametrize("use_internal", [True, False])
def test_two_independent_writers_read_alike(emu, tmp_path, use_internal):
    """tests/taxdb_writer2.py restates the constructor + serialize() a second time without sharing code with the first writer:
    compacted StringBlock (shared offsets), real Euler tour / sparse table, trailing unused bytes without internal ids, a wider M
    matrix.  The reader must produce the same taxonomy from both files (and both must agree with the dump-file loader, above)."""
    import taxdb_writer2 as tw2
    w = _world(11)
    merged = [(900001, _orig_id(5)), (900007, _orig_id(8))]
    lines, names = _lines(w)
    p1 = str(tmp_path / "w1"); p2 = str(tmp_path / "w2"); p3 = str(tmp_path / "w3")
    o2i_1 = tw.write_taxonomy_db(p1, lines, names, merged, use_internal=use_internal)
    o2i_2 = tw2.write(p2, lines, sorted(names.items()), merged, use_internal=use_internal)
    tw2.write(p3, lines, sorted(names.items()), merged, use_internal=use_internal, flog2_bias=1, seed=3)
    if use_internal:
        assert o2i_1 == o2i_2
    strains = [t for t, r in w.tax.rank.items() if r == "no rank" and t > 3]
    ids = np.array([o2i_2[_orig_id(t)] for t in strains], np.int32)
    rc1, a = _load_db(emu, p1, ids)
    rc2, b = _load_db(emu, p2, ids)
    rc3, c = _load_db(emu, p3, ids)
    assert rc1 == 0 and rc2 == 0 and rc3 == 0, (a, b, c)
    for key in ("canon", "parent", "depth", "under", "spp", "t2s", "acc", "orig"):
        assert (a[key] == b[key]).all() and (a[key] == c[key]).all(), key
    assert a["euk"] == b["euk"] == c["euk"] != 0
    assert open(p1, "rb").read() != open(p2, "rb").read()          # (different bytes: string block and tour tables)


@pytest.fixture(scope="module")
def taxonomy_check(tmp_path_factory):
    import subprocess
    exe = str(tmp_path_factory.mktemp("taxchk") / "taxonomy_check")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", exe, os.path.join(root, "tests", "emu", "taxonomy_check.cpp")])
    return exe


def test_damaged_taxonomies_are_refused(taxonomy_check, tmp_path):
    """What mtb_index_open's host side does with taxonomy files that do not describe one tree: an error, not a walk that never ends
    (the LCA walks here and on the device rely on a single root that is its own parent) and not tables sized by a garbage id.
    Found by mutating valid files under AddressSanitizer (byte flips, truncation, zeroed ranges, extreme integers: no crash)."""
    import subprocess

    def check(lines, merged=b""):
        d = tmp_path / f"t{check.n}"; check.n += 1
        os.makedirs(d)
        (d / "nodes.dmp").write_bytes(b"".join(b"%d\t|\t%d\t|\t%s\t|\t\t|\n" % (a, b, r) for a, b, r in lines))
        (d / "names.dmp").write_bytes(b"".join(b"%d\t|\tn%d\t|\t\t|\tscientific name\t|\n" % (a, a) for a, _, _ in lines) + b"3\t|\tEukaryota\t|\t\t|\tscientific name\t|\n")
        (d / "merged.dmp").write_bytes(merged)
        return subprocess.run([taxonomy_check, "dmp", str(d)], capture_output=True, timeout=60)
    check.n = 0
    good = [(1, 1, b"no rank"), (2, 1, b"superkingdom"), (3, 1, b"superkingdom"), (4, 2, b"genus"), (5, 4, b"species"), (6, 3, b"species")]
    r = check(good)
    assert r.returncode == 0 and b"ok max_id 6, 6 nodes, eukaryota 3" in r.stdout
    r = check(good[:4] + [(5, 6, b"species"), (6, 5, b"genus")])                      # two nodes that are each other's parent
    assert r.returncode == 1 and b"cycle" in r.stderr
    r = check(good + [(7, 7, b"no rank"), (8, 7, b"species")])                        # a second tree
    assert r.returncode == 1 and b"more than one root" in r.stderr
    r = check(good + [(9, 10, b"species")])
    assert r.returncode == 1 and b"missing parent" in r.stderr
    r = check(good + [(2000000000, 1, b"species")])
    assert r.returncode == 1 and b"implausible taxon id" in r.stderr
    r = check(good + [(-7, 1, b"species")])
    assert r.returncode == 1 and b"negative" in r.stderr
    r = check(good, merged=b"77\t|\t5\t|\n-3\t|\t5\t|\n5\t|\t-3\t|\n")               # aliases: a usable one and two with negative ids (ignored)
    assert r.returncode == 0 and b"ok max_id 77" in r.stdout
