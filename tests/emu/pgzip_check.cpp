// test-only: inflate a gzip file with metabuli_amd/csrc/host/pgzip.h (block-parallel) and write the text to stdout
// usage: pgzip_check FILE THREADS CHUNK_BYTES [WANT_BYTES per call]
#include <cstdio>
#include <cstdlib>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include "../../metabuli_amd/csrc/host/fastx.h"
#include "../../metabuli_amd/csrc/host/pgzip.h"
int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: pgzip_check FILE THREADS CHUNK_BYTES [WANT]\n"); return 2; }
    try {
        int fd = open(argv[1], O_RDONLY); if (fd < 0) throw std::runtime_error("cannot open");
        struct stat sb; fstat(fd, &sb);
        const uint8_t *m = (const uint8_t *)mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        const int threads = atoi(argv[2]);
        mtbhost::WorkerPool pool(threads);
        mtbhost::ParallelGzip z(m, (size_t)sb.st_size, threads, [&](size_t n, const std::function<void(size_t)> &f) { pool.run(n, f); pool.rethrow(); }, (size_t)atoll(argv[3]));
        const size_t want = argc > 4 ? (size_t)atoll(argv[4]) : (size_t)64 << 20;
        while (!z.done()) {
            mtbhost::PodVec<char> out;
            z.produce(out, want);
            fwrite(out.data(), 1, out.size(), stdout);
        }
    } catch (const std::exception &e) { fprintf(stderr, "error: %s\n", e.what()); return 1; }
    return 0;
}
