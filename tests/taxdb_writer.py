"""`taxonomyDB` as the reference's `build` writes it, restated in Python (TEST INFRASTRUCTURE ONLY): the internal
numbering of TaxonomyWrapper's constructor with useInternalTaxID = true (TaxonomyWrapper.cpp:88-113, loadNodes :147-195,
loadMerged :199-243, loadNames :245-286) and TaxonomyWrapper::serialize (:289-361).  Used to produce the fixtures the
reader in metabuli_amd/csrc/host_db.h (load_taxonomy_db) is tested on.

The MMseqs2 pieces (TaxonNode, StringBlock<unsigned int>, the Euler tour tables E/L/H/M) are absent from the reference
snapshot; their layout here follows the published MMseqs2 sources and is unpinned.  E, L, H and M are written with the
right sizes but zero content (the reader under test does not use them), so these files are fixtures for THIS repo's
reader, not input for the reference binary."""
from __future__ import annotations

import math
import struct


def internal_numbering(node_lines, merged_lines=()):
    """node_lines: [(orig_id, orig_parent, rank)] in nodes.dmp order; merged_lines: [(old, new)].
    Returns (nodes [(node_index, internal_id, internal_parent, rank)], D dict internal id -> node index,
    internal2org list, org2internal dict)."""
    org2int, int2org = {}, [0]

    def intern(o):
        if o not in org2int:
            org2int[o] = len(int2org)
            int2org.append(o)
        return org2int[o]
    nodes, dm = [], {}
    for idx, (o, op, rank) in enumerate(node_lines):
        i = intern(o)                      # the node's id first, then its parent's (loadNodes)
        ip = intern(op)
        nodes.append((idx, i, ip, rank))
        dm[i] = idx
    for old, new in merged_lines:
        io, inew = intern(old), intern(new)
        if io not in dm and inew in dm:
            dm[io] = dm[inew]
    return nodes, dm, int2org, org2int


def flog2_k(dim):
    """K = (int) MathUtil::flog2(dim) + 1; exact floor(log2) here -- the reader accepts the neighbours too"""
    return int(math.floor(math.log2(dim))) + 1


def write_taxonomy_db(path, node_lines, names: dict, merged_lines=(), use_internal=True, version=2, k_extra=0, slack=True):
    """names: orig id -> scientific name.  Returns org2internal (identity dict when use_internal is False)."""
    if use_internal:
        nodes, dm, int2org, org2int = internal_numbering(node_lines, merged_lines)
    else:
        nodes = [(idx, o, op, rank) for idx, (o, op, rank) in enumerate(node_lines)]
        dm = {o: idx for idx, (o, op, rank) in enumerate(node_lines)}
        for old, new in merged_lines:
            if old not in dm and new in dm:
                dm[old] = dm[new]
        mx = max(max(o, op) for o, op, _ in node_lines)
        mx = max([mx] + [max(a, b) for a, b in merged_lines])
        int2org = list(range(mx + 1))
        org2int = {o: o for o in range(mx + 1)}
    max_nodes = len(nodes)
    max_taxid = len(int2org) - 1
    # StringBlock<unsigned int>: ranks appended node by node (loadNodes), then the names (loadNames)
    data = bytearray()
    offsets = []

    def append(s):
        offsets.append(len(data))
        data.extend(s.encode() + b"\0")
        return len(offsets) - 1
    rank_idx = [append(rank) for (_, _, _, rank) in nodes]
    name_idx = [2**64 - 1] * max_nodes                       # (size_t)-1: no name
    node_of_internal = {i: idx for (idx, i, _, _) in nodes}
    for o, nm in names.items():
        i = org2int.get(o)
        if i is None or i not in node_of_internal:
            continue
        name_idx[node_of_internal[i]] = append(nm)
    out = bytearray()
    out += struct.pack("<i", version)
    if use_internal:
        out += struct.pack("<Q", 1)
    out += struct.pack("<Q", max_nodes)
    out += struct.pack("<i", max_taxid)
    for (idx, i, ip, _), r, n in zip(nodes, rank_idx, name_idx):
        out += struct.pack("<iii4xQQ", idx, i, ip, r, n)     # TaxonNode: 3 ints, 4 bytes padding, 2 size_t = 32 bytes
    d = [-1] * (max_taxid + 1)
    for i, idx in dm.items():
        d[i] = idx
    out += struct.pack(f"<{max_taxid + 1}i", *d)
    if use_internal:
        out += struct.pack(f"<{max_taxid + 1}i", *int2org)
    dim = 2 * max_nodes
    k = flog2_k(dim) + k_extra
    out += bytes(4 * dim) + bytes(4 * dim) + bytes(4 * max_nodes)      # E, L, H
    out += bytes(4 * dim * k)                                           # M
    out += struct.pack("<QQQ", len(data), len(offsets), len(offsets))
    out += bytes(data)
    out += struct.pack(f"<{len(offsets)}I", *offsets)
    if not use_internal and slack:
        # serialize() counts (maxTaxID + 1) ints for internal2orgTaxId into memSize whether or not it writes them
        # (TaxonomyWrapper.cpp:296-310 vs :341-344) and dumps the whole malloc'ed buffer: such files end in unused bytes
        out += b"\xCD" * (4 * (max_taxid + 1))
    with open(path, "wb") as f:
        f.write(out)
    return org2int
