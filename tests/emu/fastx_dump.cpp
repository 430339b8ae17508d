// test-only: dump what metabuli_amd/csrc/host/fastx.h parses, one "name<TAB>sequence" line per record
// (with a 5th argument "pack": the sequence decoded from the 2-bit codes + invalid mask instead of the text)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../../metabuli_amd/csrc/mtb_core.h"
#include "../../metabuli_amd/csrc/host/fastx.h"
int main(int argc, char **argv) {
    if (argc < 5) { fprintf(stderr, "usage: fastx_dump FILE THREADS BLOCK_BYTES BATCH [pack]\n"); return 2; }
    try {
        mtbhost::FastxReader r(argv[1], atoi(argv[2]), (size_t)atoll(argv[3]));
        const bool pack = argc > 5 && !strcmp(argv[5], "pack");
        mtbhost::PackTable pt;
        if (pack) {
            static mtb_tables tabs; mtb_build_tables(&tabs);
            for (int c = 0; c < 256; c++) pt.code[c] = tabs.base[c] < 4 ? tabs.base[c] : 0xFF;
            r.set_pack(&pt);
        }
        size_t batch = (size_t)atoll(argv[4]);
        for (;;) {
            mtbhost::FlatBatch b;
            if (!r.next_batch(batch, b)) break;
            uint64_t slot = 0;
            for (size_t i = 0; i < b.size(); i++) {
                fwrite(b.names.data() + b.name_offs[i], 1, b.name_offs[i + 1] - b.name_offs[i], stdout); fputc('\t', stdout);
                if (pack) {
                    const uint32_t L = b.lens[i];
                    if (L != b.offs[i + 1] - b.offs[i]) { fprintf(stderr, "error: length mismatch\n"); return 1; }
                    for (uint32_t k = 0; k < L; k++) {
                        const uint64_t g = slot + k / 8; const uint32_t j = k % 8;
                        const uint32_t w = b.packed2[2 * g] | ((uint32_t)b.packed2[2 * g + 1] << 8);
                        fputc(((b.nmask[g] >> j) & 1) ? 'N' : "ACTG"[(w >> (2 * j)) & 3], stdout);
                    }
                    slot += (L + 7) / 8;
                } else fwrite(b.bases.data() + b.offs[i], 1, b.offs[i + 1] - b.offs[i], stdout);
                fputc('\n', stdout);
            }
            if (pack && slot != b.slots) { fprintf(stderr, "error: slot count mismatch\n"); return 1; }
        }
    } catch (const std::exception &e) { fprintf(stderr, "error: %s\n", e.what()); return 1; }
    return 0;
}
