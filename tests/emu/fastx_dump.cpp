// test-only: dump what metabuli_amd/csrc/host/fastx.h parses, one "name<TAB>sequence" line per record
#include <cstdio>
#include <cstdlib>
#include "../../metabuli_amd/csrc/host/fastx.h"
int main(int argc, char **argv) {
    if (argc < 5) { fprintf(stderr, "usage: fastx_dump FILE THREADS BLOCK_BYTES BATCH\n"); return 2; }
    try {
        mtbhost::FastxReader r(argv[1], atoi(argv[2]), (size_t)atoll(argv[3]));
        size_t batch = (size_t)atoll(argv[4]);
        for (;;) {
            mtbhost::FlatBatch b;
            if (!r.next_batch(batch, b)) break;
            for (size_t i = 0; i < b.size(); i++) {
                fwrite(b.names.data() + b.name_offs[i], 1, b.name_offs[i + 1] - b.name_offs[i], stdout); fputc('\t', stdout);
                fwrite(b.bases.data() + b.offs[i], 1, b.offs[i + 1] - b.offs[i], stdout); fputc('\n', stdout);
            }
        }
    } catch (const std::exception &e) { fprintf(stderr, "error: %s\n", e.what()); return 1; }
    return 0;
}
