// test-only: the greedy path combination (Taxonomer::combineMatchPaths, Taxonomer.cpp:428-468; trimMatchPath :475-485) evaluated
// 64 candidates at a time must give what the one-candidate-at-a-time loop gives -- same accepted paths, same trims, the species
// score bit for bit.  This is the formulation a wavefront can run (a lane per candidate; an accepted path's ends are broadcast):
//   per batch: every lane runs its candidate against the paths accepted BEFORE the batch, in order; then the first surviving lane
//   is accepted (nothing accepted later can precede it), the lanes behind it run against that path, and so on.
// A candidate therefore meets exactly the accepted paths it meets in the serial loop, in the same order.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <random>
#include <vector>
#include "../../metabuli_amd/csrc/mtb_core.h"
struct P { int32_t start, end; float score; int32_t ham; uint32_t reh_start, reh_end; };
static bool against(P &p, int32_t cst, int32_t cen) {          // returns true if the candidate is dropped
    if (!((p.end < cst) || (cen < p.start))) {
        const int32_t ov = (p.end < cen ? p.end : cen) - (p.start > cst ? p.start : cst) + 1;
        if (ov == p.end - p.start + 1) return true;
        if (ov < 24) {
            if (p.start < cst) {
                p.end = cst - 1;
                const int32_t h = p.ham - mtb_part_ham(p.reh_end, ov / 3, false); p.ham = h > 0 ? h : 0;
                p.score = p.score - mtb_part_score(p.reh_end, ov / 3, false) - (float)(ov % 3);
            } else {
                p.start = cen + 1;
                const int32_t h = p.ham - mtb_part_ham(p.reh_start, ov / 3, true); p.ham = h > 0 ? h : 0;
                p.score = p.score - mtb_part_score(p.reh_start, ov / 3, true) - (float)(ov % 3);
            }
        } else return true;
    }
    return false;
}
static float serial(std::vector<P> c, std::vector<P> *acc) {
    float score = 0.0f;
    for (size_t k = 0; k < c.size(); k++) {
        P p = c[k]; bool drop = false;
        for (size_t a = 0; a < acc->size() && !drop; a++) drop = against(p, (*acc)[a].start, (*acc)[a].end);
        if (!drop) { acc->push_back(p); score += p.score; }
    }
    return score;
}
static float batched(const std::vector<P> &c, std::vector<P> *acc, int W) {
    float score = 0.0f;
    for (size_t b0 = 0; b0 < c.size(); b0 += (size_t)W) {
        const size_t nb = std::min<size_t>((size_t)W, c.size() - b0);
        std::vector<P> lane(c.begin() + (long)b0, c.begin() + (long)(b0 + nb));
        std::vector<char> drop(nb, 0), taken(nb, 0);
        const size_t na0 = acc->size();
        for (size_t a = 0; a < na0; a++)                                    // phase 1: the paths accepted before the batch, in order
            for (size_t l = 0; l < nb; l++) if (!drop[l]) drop[l] = against(lane[l], (*acc)[a].start, (*acc)[a].end);
        for (;;) {                                                           // phase 2: survivors in lane order
            size_t f = nb;
            for (size_t l = 0; l < nb; l++) if (!drop[l] && !taken[l]) { f = l; break; }
            if (f == nb) break;
            taken[f] = 1; acc->push_back(lane[f]); score += lane[f].score;
            for (size_t l = f + 1; l < nb; l++) if (!drop[l]) drop[l] = against(lane[l], lane[f].start, lane[f].end);
        }
    }
    return score;
}
int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 20000;
    std::mt19937_64 r(7);
    for (int it = 0; it < rounds; it++) {
        const int n = 1 + (int)(r() % 300), span = 200 + (int)(r() % 10000);
        std::vector<P> c((size_t)n);
        for (auto &p : c) {
            p.start = (int32_t)(r() % (uint64_t)span); p.end = p.start + 23 + 3 * (int32_t)(r() % 40);
            p.score = 0.5f * (float)(6 + r() % 200); p.ham = (int32_t)(r() % 12); p.reh_start = (uint32_t)(r() & 0xFFFF); p.reh_end = (uint32_t)(r() & 0xFFFF);
        }
        std::stable_sort(c.begin(), c.end(), [](const P &a, const P &b) { if (a.score != b.score) return a.score > b.score; if (a.ham != b.ham) return a.ham < b.ham; return a.start > b.start; });
        std::vector<P> a1, a2, a3;
        const float s1 = serial(c, &a1), s2 = batched(c, &a2, 64), s3 = batched(c, &a3, 7);
        if (memcmp(&s1, &s2, 4) || memcmp(&s1, &s3, 4) || a1.size() != a2.size() || a1.size() != a3.size()) { printf("MISMATCH round %d: %g %g %g, %zu %zu %zu\n", it, s1, s2, s3, a1.size(), a2.size(), a3.size()); return 1; }
        for (size_t k = 0; k < a1.size(); k++)
            if (memcmp(&a1[k], &a2[k], sizeof(P)) || memcmp(&a1[k], &a3[k], sizeof(P))) { printf("MISMATCH round %d path %zu\n", it, k); return 1; }
    }
    printf("OK %d\n", rounds);
    return 0;
}
