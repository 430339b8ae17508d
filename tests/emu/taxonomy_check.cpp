// test-only: load a taxonomy with metabuli_amd/csrc/host_db.h (what mtb_index_open does on the host) and print a summary
// usage: taxonomy_check dmp DIR | db FILE | list FILE | params DIR
#include <cstdio>
#include <cstring>
#include "../../include/mtb.h"
#include "../../metabuli_amd/csrc/host_db.h"
int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: taxonomy_check dmp DIR | db FILE | list FILE | params DIR\n"); return 2; }
    std::string err;
    if (!strcmp(argv[1], "list")) { std::vector<int32_t> ids; if (!mtbhost::read_taxid_list(argv[2], &ids)) { fprintf(stderr, "error: cannot open\n"); return 1; } printf("ok %zu ids\n", ids.size()); return 0; }
    if (!strcmp(argv[1], "params")) { mtb_params p; memset(&p, 0, sizeof(p)); int r = 0; mtbhost::load_db_parameters(argv[2], &p, &r); printf("ok %d %d %d %d\n", p.syncmer, p.smer_len, p.kmer_format, r); return 0; }
    mtbhost::Taxonomy t;
    const bool ok = !strcmp(argv[1], "db") ? mtbhost::load_taxonomy_db(argv[2], &t, &err) : mtbhost::load_taxonomy(argv[2], &t, &err);
    if (!ok) { fprintf(stderr, "error: %s\n", err.c_str()); return 1; }
    std::vector<int32_t> ids;
    for (int32_t i = 0; i <= t.max_id; i++) if (t.cn(i) == i) ids.push_back(i);
    mtbhost::build_tax2species(&t, ids.data(), ids.size());
    long long acc = 0;
    for (size_t i = 0; i + 1 < ids.size(); i += 1 + ids.size() / 64) acc += t.lca(ids[i], ids[i + 1]) + (long long)t.children_of(ids[i]).size();
    printf("ok max_id %d, %zu nodes, eukaryota %d, checksum %lld\n", t.max_id, ids.size(), t.eukaryota, acc);
    return 0;
}
