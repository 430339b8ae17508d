"""Reporter of `metabuli classify`, restated in Python straight from src/commons/Reporter.cpp (TEST INFRASTRUCTURE ONLY).

The driver's TSV writers (metabuli_amd/csrc/host/classify_main.cpp) are checked against THIS file, not against the
C++ twin inside oracle/oracle.cpp: written in another language, from the reference text, with the reference's data
flow (getParentToChildren + getCladeCounts, then the recursive writer).

  write_classifications   Reporter::writeReadClassification  Reporter.cpp:35-80
  write_report            Reporter::writeReportFile / writeReport  Reporter.cpp:115-193
  krona_nodes             Reporter::kronaReport  Reporter.cpp:86-113 (the <node> tree between the HTML prelude and tail)
  lineage                 TaxonomyWrapper::taxLineage2  TaxonomyWrapper.cpp:431-454 (short ranks: TaxonomyWrapper.h:9-26)
"""
from __future__ import annotations

SHORT_RANKS = {"subspecies": "ss", "species": "s", "subgenus": "sg", "genus": "g", "subfamily": "sf", "family": "f", "suborder": "so",
               "order": "o", "subclass": "sc", "class": "c", "subphylum": "sp", "phylum": "p", "subkingdom": "sk", "kingdom": "k",
               "superkingdom": "d", "domain": "d", "realm": "r"}


def ostream_float(x) -> str:
    """`ostream << float` with the default precision (6 significant digits, %g)"""
    return "%g" % float(x)


class TaxView:
    """what Reporter needs from TaxonomyWrapper: parent / rank / name by (internal) id and getOriginalTaxID"""

    def __init__(self, parent: dict, rank: dict, name: dict, orig=None):
        self.parent, self.rank, self.name = parent, rank, name
        self.orig = orig

    def original(self, t):
        return t if self.orig is None else self.orig[t]

    def lineage(self, t):
        chain = []
        node = t
        while True:                                  # do { push; node = parent } while (node is not the root)
            chain.append(node)
            node = self.parent[node]
            if self.parent[node] == node:
                break
        return ";".join(f"{SHORT_RANKS.get(self.rank[n], '-')}_{self.name[n]}" for n in reversed(chain))


def write_classifications(path, tax: TaxView, names, results, tc_tax, tc_cnt, lineage=False):
    with open(path, "w") as f:
        f.write("#is_classified\tname\ttaxID\tquery_length\tscore\trank")
        if lineage:
            f.write("\tlineage")
        f.write("\ttaxID:match_count\n")
        for i, r in enumerate(results):
            cls = int(r["classification"])
            f.write(f"{int(r['is_classified'])}\t{names[i]}\t{tax.original(cls)}\t{int(r['qlen']) + int(r['qlen2'])}\t{ostream_float(r['score'])}\t")
            if r["is_classified"]:
                f.write(tax.rank[cls] + "\t")
                if lineage:
                    f.write(tax.lineage(cls) + "\t")
                a = int(r["taxcnt_off"])
                for k in range(a, a + int(r["n_taxcnt"])):       # std::map order = ascending internal id
                    f.write(f"{tax.original(int(tc_tax[k]))}:{int(tc_cnt[k])} ")
                f.write("\n")
            else:
                f.write("-\t")
                if lineage:
                    f.write("-\t")
                f.write("-\t\n")


def clade_counts(tax: TaxView, tax_counts: dict):
    """NcbiTaxonomy::getCladeCounts as used at Reporter.cpp:121-122: every counted taxon adds its count to itself and to all
    of its ancestors; a node's children list is the taxonomy's (all children, counted or not)."""
    clade, own = {}, {}
    for t, c in tax_counts.items():
        own[t] = c
        clade[t] = clade.get(t, 0) + c
        if t == 0:
            continue
        node = t
        while tax.parent[node] != node:
            node = tax.parent[node]
            clade[node] = clade.get(node, 0) + c
    children = {}
    for t, p in tax.parent.items():
        if p != t:
            children.setdefault(p, []).append(t)
    return clade, own, children


def _walk(tax, clade, own, children, total, t, depth, emit):
    c = clade.get(t, 0)
    if c == 0:
        return
    emit(t, depth, c, own.get(t, 0), True)
    for ch in sorted(children.get(t, []), key=lambda x: -clade.get(x, 0)):     # SORT_SERIAL by clade count, descending
        if ch in clade:
            _walk(tax, clade, own, children, total, ch, depth + 1, emit)
        else:
            break
    emit(t, depth, c, own.get(t, 0), False)


def write_report(path, tax: TaxView, tax_counts: dict, total: int):
    clade, own, children = clade_counts(tax, tax_counts)
    with open(path, "w") as f:
        f.write("#clade_proportion\tclade_count\ttaxon_count\trank\ttaxID\tname\n")
        if clade.get(0, 0) > 0:
            f.write("%.4f\t%i\t%i\tno rank\t0\tunclassified\n" % (100 * clade[0] / float(total), clade[0], own.get(0, 0)))

        def emit(t, depth, c, o, opening):
            if opening:
                f.write("%.4f\t%i\t%i\t%s\t%i\t%s%s\n" % (100 * c / float(total), c, o, tax.rank[t], tax.original(t), " " * (2 * depth), tax.name[t]))
        _walk(tax, clade, own, children, total, 1, 0, emit)


def escape_attribute(s: str) -> str:
    """escapeAttribute of the krona writer: the five XML specials"""
    return s.replace("&", "&amp;").replace('"', "&quot;").replace("'", "&apos;").replace("<", "&lt;").replace(">", "&gt;")


def krona_nodes(tax: TaxView, tax_counts: dict, total: int) -> str:
    """the XML between krona_prelude_html and "</krona></div></body></html>" (Reporter.cpp:155-158)"""
    clade, own, children = clade_counts(tax, tax_counts)
    out = ['<node name="all"><magnitude><val>%d</val></magnitude>' % total]
    if clade.get(0, 0) > 0:
        out.append('<node name="unclassified"><magnitude><val>%d</val></magnitude></node>' % clade[0])

    def emit(t, depth, c, o, opening):
        out.append('<node name="%s"><magnitude><val>%d</val></magnitude>' % (escape_attribute(tax.name[t]), c) if opening else "</node>")
    _walk(tax, clade, own, children, total, 1, 0, emit)
    out.append("</node>")
    return "".join(out)
