/* kernels_index.h -- index residency.
 *
 * (1) One-time GPU decode of the delta-coded target list `diffIdx` into a flat
 *     u64 value[T] array (the reference decodes it sequentially inside every
 *     matchKmers call: KmerMatcher::getNextTargetKmer, KmerMatcher.h:282-297;
 *     format written by IndexCreator::getDiffIdx, IndexCreator.cpp:874-892):
 *     terminator flags -> tile counts -> scan -> per-metamer delta assembly
 *     (big-endian 15-bit groups) -> 64-bit inclusive prefix sum.
 * (2) Synthetic "GTDB-scale" filler index generated directly in sorted order
 *     (SURVEY.md 8(d)): entry i is a pure function of (seed, i), strictly
 *     monotone in its amino-acid part, so no sort is needed and the real
 *     (genome-derived) entries are merged in with two rank searches.        */
#ifndef MTB_KERNELS_INDEX_H
#define MTB_KERNELS_INDEX_H
#include "dev_util.h"
#include "kernels_scan.h"
#include "mtb_core.h"

/* ---------------- diffIdx decode ---------------- */
__global__ __launch_bounds__(256) void k_diff_tile_count(const uint16_t *__restrict__ d, uint64_t n16, uint32_t *__restrict__ tile_cnt) {
    __shared__ uint32_t s_tmp[8];
    uint64_t base = (uint64_t)blockIdx.x * 2048 + (uint64_t)threadIdx.x * 8;
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) { uint64_t i = base + k; if (i < n16) c += d[i] >> 15; }
    uint32_t tot;
    block256_exclusive_scan<uint32_t>(c, s_tmp, &tot);
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void k_diff_assemble(const uint16_t *__restrict__ d, uint64_t n16,
                                                        const uint64_t *__restrict__ tile_off, uint64_t *__restrict__ deltas) {
    __shared__ uint32_t s_tmp[8];
    uint64_t base = (uint64_t)blockIdx.x * 2048 + (uint64_t)threadIdx.x * 8;
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) { uint64_t i = base + k; if (i < n16) c += d[i] >> 15; }
    uint32_t tot;
    uint64_t o = tile_off[blockIdx.x] + block256_exclusive_scan<uint32_t>(c, s_tmp, &tot);
    for (int k = 0; k < 8; k++) {
        uint64_t j = base + k;
        if (j >= n16) break;
        uint16_t f = d[j];
        if (!(f & 0x8000u)) continue;
        uint64_t dl = f & 0x7FFFu;
        int sh = 15;
        uint64_t q = j;
        while (q > 0 && sh < 75) {
            uint16_t g = d[q - 1];
            if (g & 0x8000u) break;
            if (sh < 64) dl |= (uint64_t)g << sh;
            sh += 15; q--;
        }
        deltas[o++] = dl;
    }
}

/* first delta of a chunk += the last value of the previous chunk (chunked decode: the 64-bit prefix sum then runs per chunk);
 * k_save_last keeps the chunk's last FLAT value for the next chunk (the array may be packed in place right afterwards) */
__global__ void k_diff_add_carry(uint64_t *__restrict__ first_delta, const uint64_t *__restrict__ carry) { *first_delta += *carry; }
__global__ void k_save_last(const uint64_t *__restrict__ last, uint64_t *__restrict__ carry) { *carry = *last; }

/* ---------------- diffIdx encode (mtb_index_write): IndexCreator::getDiffIdx (IndexCreator.cpp:874-892) on the device ----------------
 * words of entry i = 15-bit groups of (value[i] - value[i-1]), most significant first, the last one flagged 0x8000 */
__device__ __forceinline__ uint32_t mtb_diff_words(uint64_t dlt) {
    const uint32_t bits = dlt ? 64u - (uint32_t)__clzll((unsigned long long)dlt) : 1u;
    return (bits + 14u) / 15u;
}
__global__ __launch_bounds__(256) void k_diff_nwords(const uint64_t *__restrict__ values, uint64_t i0, uint64_t m, uint32_t *__restrict__ nw) {
    const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    const uint64_t i = i0 + j;
    nw[j] = mtb_diff_words(values[i] - (i ? values[i - 1] : 0ull));
}
__global__ __launch_bounds__(256) void k_diff_encode(const uint64_t *__restrict__ values, uint64_t i0, uint64_t m, const uint64_t *__restrict__ off, uint16_t *__restrict__ enc) {
    const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    const uint64_t i = i0 + j;
    const uint64_t dlt = values[i] - (i ? values[i - 1] : 0ull);
    const uint32_t n = mtb_diff_words(dlt);
    uint16_t *o = enc + off[j];
    for (uint32_t q = 0; q < n; q++) {
        const uint32_t g = n - 1 - q;
        o[q] = (uint16_t)(((dlt >> (15 * g)) & 0x7FFFull) | (g == 0 ? 0x8000u : 0u));
    }
}
/* taxID_list (IndexCreator.cpp:329-333): ids that occur, as a byte map; ids outside [0, max_id] are appended to a short list */
__global__ __launch_bounds__(256) void k_mark_taxids(const uint32_t *__restrict__ info, uint64_t i0, uint64_t m, uint8_t *__restrict__ seen, int32_t max_id,
                                                      int32_t *__restrict__ extra, uint32_t extra_cap, uint32_t *__restrict__ n_extra) {
    const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    const int32_t t = (int32_t)(info[i0 + j] & 0x7FFFFFFFu);
    if (t >= 0 && t <= max_id + 1) { if (!seen[t]) seen[t] = 1; }
    else { const uint32_t at = atomicAdd(n_extra, 1u); if (at < extra_cap) extra[at] = t; }
}
/* split checkpoints (IndexCreator.cpp:848-857): checkpoint k is armed at entry k * size_of_split - 1 with that entry's amino-acid part and
 * recorded at the first later entry of another amino-acid part: j[k] = its index (n if there is none) */
__global__ __launch_bounds__(256) void k_split_find(const uint64_t *__restrict__ values, uint64_t n, uint64_t size_of_split, uint32_t n_k, uint64_t *__restrict__ j_out) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x + 1;
    if (k > n_k) return;
    const uint64_t i0 = (uint64_t)k * size_of_split - 1;
    uint64_t j = n;
    if (i0 < n) {
        const uint64_t aa = values[i0] & ~0xFFFFFFull;
        j = i0 + 1;
        while (j < n && (values[j] & ~0xFFFFFFull) == aa) j++;
    }
    j_out[k - 1] = j;
}
/* value[j] and the word offset BEHIND entry j (inside the current slice [i0, i0 + m)) for a short list of entries */
__global__ void k_split_gather(const uint64_t *__restrict__ values, const uint64_t *__restrict__ off, uint64_t i0, const uint64_t *__restrict__ js, uint32_t n,
                               uint64_t *__restrict__ out_value, uint64_t *__restrict__ out_off) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint64_t j = js[t];
    out_value[t] = values[j]; out_off[t] = off[j - i0 + 1];
}

/* ---------------- synthetic filler index ---------------- */
#define MTB_AA_SPACE 37822859361ull   /* 21^8 */

struct FillerParams {
    uint64_t seed, n_filler, stride;
    int32_t tax_lo; uint32_t tax_span;
    uint8_t ncid[21]; uint8_t cids[21][6];
};

__host__ __device__ __forceinline__ uint64_t mtb_mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__host__ __device__ __forceinline__ void filler_cand(const FillerParams &P, uint64_t idx, uint64_t aa_rank, uint64_t *value, int32_t *tax) {
    uint64_t h = mtb_mix64(P.seed * 0xD6E8FEB86659FD93ull + idx);
    uint64_t aa = 0, dna = 0;
    uint32_t dig[8];
    uint64_t r = aa_rank;
    for (int k = 7; k >= 0; k--) { dig[k] = (uint32_t)(r % 21); r /= 21; }
    uint64_t hh = h;
    for (int k = 0; k < 8; k++) {
        aa = (aa << 5) | dig[k];
        uint32_t nc = P.ncid[dig[k]];
        dna = (dna << 3) | P.cids[dig[k]][(hh & 0xFF) % nc];
        hh >>= 5;
    }
    *value = (aa << 24) | dna;
    *tax = P.tax_lo + (int32_t)((h >> 40) % P.tax_span);
}

__host__ __device__ __forceinline__ void filler_entry(const FillerParams &P, uint64_t i, uint64_t *value, int32_t *tax) {
    uint64_t g = i >> 1;
    uint64_t h = mtb_mix64(P.seed ^ mtb_mix64(g));
    uint64_t base = g * P.stride + h % (P.stride - 1);
    bool shared = ((h >> 40) & 3u) == 0;
    if (!shared) { filler_cand(P, i, base + (i & 1), value, tax); return; }
    uint64_t v0, v1; int32_t t0, t1;
    filler_cand(P, 2 * g, base, &v0, &t0);
    filler_cand(P, 2 * g + 1, base, &v1, &t1);
    if (v1 < v0 || (v1 == v0 && t1 < t0)) { uint64_t tv = v0; v0 = v1; v1 = tv; int32_t tt = t0; t0 = t1; t1 = tt; }
    if (v0 == v1 && t0 == t1) {
        if (P.tax_span > 1) { if (t0 == P.tax_lo + (int32_t)P.tax_span - 1) t0 -= 1; else t1 += 1; }
    }
    if (i & 1) { *value = v1; *tax = t1; } else { *value = v0; *tax = t0; }
}

/* position of every real entry: its rank + number of fillers with a smaller value */
__global__ void k_synth_real_pos(FillerParams P, const uint64_t *__restrict__ rv, uint64_t n_real, uint64_t *__restrict__ pos) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_real) return;
    uint64_t v = rv[j];
    uint64_t lo = 0, hi = P.n_filler;
    while (lo < hi) {
        uint64_t mid = lo + ((hi - lo) >> 1);
        uint64_t fv; int32_t ft;
        filler_entry(P, mid, &fv, &ft);
        if (fv < v) lo = mid + 1; else hi = mid;
    }
    pos[j] = j + lo;
}

/* grid-stride over 256-entry chunks: an AQL dispatch holds at most 2^32-1
 * work-items, fewer than the entries of a GTDB-scale index */
__global__ __launch_bounds__(256) void k_synth_fill(FillerParams P, const uint64_t *__restrict__ rv, uint64_t n_real,
                                                     uint64_t *__restrict__ values, uint32_t *__restrict__ info) {
    __shared__ uint64_t s_r0;
    const uint64_t chunks = (P.n_filler + 255) / 256;
    for (uint64_t ch = blockIdx.x; ch < chunks; ch += gridDim.x) {
        uint64_t i0 = ch * 256;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint64_t fv; int32_t ft;
            filler_entry(P, i0, &fv, &ft);
            uint64_t lo = 0, hi = n_real;           /* reals with value <= fv */
            while (lo < hi) { uint64_t mid = lo + ((hi - lo) >> 1); if (rv[mid] <= fv) lo = mid + 1; else hi = mid; }
            s_r0 = lo;
        }
        __syncthreads();
        uint64_t i = i0 + threadIdx.x;
        if (i >= P.n_filler) continue;
        uint64_t v; int32_t t;
        filler_entry(P, i, &v, &t);
        uint64_t r = s_r0;
        while (r < n_real && rv[r] <= v) r++;
        values[i + r] = v;
        info[i + r] = (uint32_t)t;
    }
}

__global__ void k_synth_place_real(const uint64_t *__restrict__ rv, const int32_t *__restrict__ rt, const uint64_t *__restrict__ pos,
                                   uint64_t n_real, uint64_t *__restrict__ values, uint32_t *__restrict__ info) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_real) return;
    values[pos[j]] = rv[j];
    info[pos[j]] = (uint32_t)rt[j];
}

#endif
