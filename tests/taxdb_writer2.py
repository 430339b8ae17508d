"""A SECOND writer of the binary `taxonomyDB` (TEST INFRASTRUCTURE ONLY), written from TaxonomyWrapper's constructor and
TaxonomyWrapper::serialize line by line (src/commons/TaxonomyWrapper.cpp:66-112 constructor, :147-195 loadNodes, :199-243 loadMerged,
:245-286 loadNames, :113-145 initTaxonomy, :289-361 serialize) and sharing NO helper with tests/taxdb_writer.py, so that the reader
(metabuli_amd/csrc/host_db.h::load_taxonomy_db) and its first fixture writer stop sharing assumptions:

  * the tables are numpy arrays written with tobytes() (the first writer struct.pack's field by field);
  * the StringBlock is COMPACTED as StringBlock::compact() does before serialisation (sorted unique strings, equal strings share
    one offset: every "species" rank points at the same bytes) -- the first writer appends one copy per node;
  * E / L / H and the sparse table M hold the real Euler tour and range-minimum table (NcbiTaxonomy::elh, computeSparseTable;
    the first writer zero-fills them);
  * without internal ids the file ends in the (maxTaxID + 1) unused ints serialize() always reserves (filled with garbage).

The MMseqs2 pieces (TaxonNode, StringBlock, flog2) are not in the reference snapshot: their layout follows the published MMseqs2
sources and stays unpinned -- two writers agreeing is evidence about the reader, not about the reference's bytes."""
import io
import math

import numpy as np

NODE_DT = np.dtype([("id", "<i4"), ("taxId", "<i4"), ("parentTaxId", "<i4"), ("_pad", "<i4"), ("rankIdx", "<u8"), ("nameIdx", "<u8")])
assert NODE_DT.itemsize == 32          # sizeof(TaxonNode): 3 ints, 4 bytes of padding, 2 size_t


class _Block:
    """StringBlock<unsigned int>: append() returns the entry index; compact() + serialize() as in MMseqs2's StringBlock.h"""

    def __init__(self):
        self.entries = []

    def append(self, s):
        self.entries.append(s.encode())
        return len(self.entries) - 1

    def serialized(self):
        order = sorted(range(len(self.entries)), key=lambda i: self.entries[i])          # compact(): entries sorted by their string
        data = io.BytesIO()
        offsets = np.zeros(len(self.entries), "<u4")
        prev = None
        for i in order:
            if prev is not None and self.entries[i] == self.entries[prev]:
                offsets[i] = offsets[prev]                                                # equal strings share their bytes
            else:
                offsets[i] = data.tell()
                data.write(self.entries[i] + b"\0")
            prev = i
        raw = data.getvalue()
        head = np.array([len(raw), len(self.entries), len(self.entries)], "<u8")         # byteCapacity, entryCapacity, entryCount
        return head.tobytes() + raw + offsets.tobytes()


def write(path, nodes_dmp, names_dmp, merged_dmp=(), use_internal=True, flog2_bias=0, seed=0):
    """nodes_dmp: [(taxid, parent, rank)] in file order; names_dmp: [(taxid, scientific name)]; merged_dmp: [(old, new)].
    Returns the original -> internal id map (identity without internal ids)."""
    block = _Block()
    o2i, i2o = {}, [0]
    nodes = []                                   # (internal id, internal parent, rank entry)
    dm = {}                                      # internal taxid -> node index
    if use_internal:
        def internal(o):
            if o not in o2i:
                o2i[o] = len(i2o); i2o.append(o)
            return o2i[o]
        for org, parent, rank in nodes_dmp:      # loadNodes: the node's id is numbered before its parent's
            t = internal(org); p = internal(parent)
            dm[t] = len(nodes)
            nodes.append((t, p, block.append(rank)))
        for old, new in merged_dmp:              # loadMerged: both ids are numbered, the alias only if the old one has no node
            a = internal(old); b = internal(new)
            if a not in dm and b in dm:
                dm[a] = dm[b]
        max_taxid = len(i2o) - 1
    else:
        for org, parent, rank in nodes_dmp:
            dm[org] = len(nodes)
            nodes.append((org, parent, block.append(rank)))
        for old, new in merged_dmp:
            if old not in dm and new in dm:
                dm[old] = dm[new]
        max_taxid = max(dm)
        o2i = {t: t for t in range(max_taxid + 1)}
    n = len(nodes)
    tab = np.zeros(n, NODE_DT)
    tab["id"] = np.arange(n); tab["taxId"] = [x[0] for x in nodes]; tab["parentTaxId"] = [x[1] for x in nodes]
    tab["rankIdx"] = [x[2] for x in nodes]; tab["nameIdx"] = np.uint64(2**64 - 1)         # (size_t)-1 until a name arrives
    for org, name in names_dmp:                  # loadNames
        t = o2i[org] if use_internal else org
        tab["nameIdx"][dm[t]] = block.append(name)
    D = np.full(max_taxid + 1, -1, "<i4")
    for t, idx in dm.items():
        D[t] = idx
    # initTaxonomy: Euler tour from taxid 1 (E = node indices, L = levels, H = first occurrence), sparse table of level minima
    children = [[] for _ in range(n)]
    for t, p, _ in nodes:
        if p != t:
            children[dm[p]].append(t)
    E, L = [], []
    H = np.zeros(n, "<i4")
    stack = [(1, 0, 0)]                          # (taxid, level, next child): elh() without recursion
    while stack:
        t, lvl, k = stack.pop()
        idx = dm[t]
        if k == 0 and H[idx] == 0:
            H[idx] = len(E)
        E.append(idx); L.append(lvl)
        if k < len(children[idx]):
            stack.append((t, lvl, k + 1))
            stack.append((children[idx][k], lvl + 1, 0))
    dim = 2 * n
    Ea = np.zeros(dim, "<i4"); La = np.zeros(dim, "<i4")
    Ea[:len(E)] = E; La[:len(L)] = L
    K = int(math.floor(math.log2(dim))) + 1 + flog2_bias          # (int) MathUtil::flog2(dim) + 1; flog2 is approximate
    M = np.zeros((dim, K), "<i4")
    M[:, 0] = np.arange(dim)
    col = 1
    while (1 << col) <= dim and col < K:
        rows = np.arange(0, dim - (1 << col) + 1)
        a = M[rows, col - 1]; b = M[rows + (1 << (col - 1)), col - 1]
        M[rows, col] = np.where(La[a] < La[b], a, b)
        col += 1
    out = io.BytesIO()
    out.write(np.array([2], "<i4").tobytes())                     # SERIALIZATION_VERSION
    if use_internal:
        out.write(np.array([1], "<u8").tobytes())                 # internalTaxIdUsed
    out.write(np.array([n], "<u8").tobytes())
    out.write(np.array([max_taxid], "<i4").tobytes())
    out.write(tab.tobytes())
    out.write(D.tobytes())
    if use_internal:
        out.write(np.array(i2o, "<i4").tobytes())
    out.write(Ea.tobytes()); out.write(La.tobytes()); out.write(H.tobytes()); out.write(M.tobytes())
    out.write(block.serialized())
    if not use_internal:                                           # memSize counts the id map although it is not written
        out.write(np.random.default_rng(seed).integers(0, 256, 4 * (max_taxid + 1), dtype=np.uint8).tobytes())
    with open(path, "wb") as f:
        f.write(out.getvalue())
    return o2i
