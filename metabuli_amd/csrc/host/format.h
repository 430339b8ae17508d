/* format.h -- text output helpers of the host driver: a growing byte buffer written through a raw cursor, integers, and a float
 * printed the way the reference's `ostream << float` prints it (Reporter.cpp:50: default floatfield, precision 6 = printf("%g")).
 * The formatter prints three numbers per read plus the score; through snprintf / std::string::operator+= that was the stage's
 * cost (35 M reads/s on 32 threads).  Host code only; no device dependency. */
#ifndef MTB_HOST_FORMAT_H
#define MTB_HOST_FORMAT_H
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

namespace mtbhost {

/* append-only buffer over a std::string: reserve() once per row, then raw stores */
struct RowBuf {
    std::string &s; char *p; char *end;
    explicit RowBuf(std::string &str, size_t first) : s(str) { s.resize(first < 256 ? 256 : first); p = &s[0]; end = p + s.size(); }
    inline void need(size_t n) {
        if ((size_t)(end - p) >= n) return;
        const size_t used = (size_t)(p - s.data());
        s.resize((used + n) * 2);
        p = &s[0] + used; end = &s[0] + s.size();
    }
    inline void ch(char c) { *p++ = c; }
    inline void bytes(const char *b, size_t n) { memcpy(p, b, n); p += n; }
    inline void cstr(const char *b) { const size_t n = strlen(b); need(n + 64); bytes(b, n); }
    void finish() { s.resize((size_t)(p - s.data())); }
};

/* decimal digits of v at p; returns the end (at most 20 characters, 21 with the sign) */
inline char *put_uint(char *p, unsigned long long u) {
    char buf[24]; int n = 0;
    do { buf[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    while (n) *p++ = buf[--n];
    return p;
}
inline char *put_int(char *p, long long v) {
    if (v < 0) { *p++ = '-'; return put_uint(p, (unsigned long long)(-(v + 1)) + 1ull); }
    return put_uint(p, (unsigned long long)v);
}

/* printf("%g", (double)v) for a float, at most 16 characters.  Exact: the float is m * 2^-s with a 24-bit m; for 1e-4 <= v < 1e6 the six
 * significant digits are round-half-even(m * 10^k / 2^s) in 64-bit integer arithmetic (m * 10^9 < 2^54), the decimal exponent is
 * found from the quotient itself, trailing zeros are dropped as %g drops them.  Everything else (0 handled, negatives, tiny, huge, inf,
 * nan) goes through snprintf. */
inline char *put_float_g6(char *p, float v) {
    if (v == 0.0f && !std::signbit(v)) { *p++ = '0'; return p; }
    uint32_t bits; memcpy(&bits, &v, 4);
    const uint32_t ex = (bits >> 23) & 0xFFu;
    if (!(v >= 1e-4f && v < 1e6f) || ex == 0) { return p + snprintf(p, 32, "%g", (double)v); }
    const uint64_t m = (bits & 0x7FFFFFu) | 0x800000u;
    const int s = 150 - (int)ex;                                  /* v = m * 2^-s; v < 1e6 < 2^20 -> s >= 4; v >= 1e-4 -> s <= 37 */
    static const uint64_t P10[11] = {1ull, 10ull, 100ull, 1000ull, 10000ull, 100000ull, 1000000ull, 10000000ull, 100000000ull, 1000000000ull, 10000000000ull};
    /* X = floor(log10 v): a guess from the binary exponent, corrected on the quotient (six digits <=> 1e5 <= q < 1e6) */
    int X = (((int)ex - 127) * 1233) / 4096;
    if (X < -5) X = -5;
    if (X > 5) X = 5;
    uint64_t q, rem;
    for (;;) {
        const uint64_t N = m * P10[5 - X];                         /* 5 - X = 0 .. 10: N < 2^24 * 10^10 < 2^58 */
        q = N >> s; rem = N & ((1ull << s) - 1ull);
        if (q >= 1000000ull && X < 5) { X++; continue; }
        if (q < 100000ull && X > -5) { X--; continue; }
        break;
    }
    const uint64_t half = 1ull << (s - 1);
    if (rem > half || (rem == half && (q & 1ull))) q++;
    if (q >= 1000000ull) { q = 100000ull; X++; }
    if (X >= 6 || X < -4 || q < 100000ull) return p + snprintf(p, 32, "%g", (double)v);      /* exponent form (rounded up to 1e+06, or below 1e-04) */
    char d[6];
    for (int i = 5; i >= 0; i--) { d[i] = (char)('0' + q % 10); q /= 10; }
    int nd = 6;
    while (nd > 1 && d[nd - 1] == '0') nd--;                      /* %g drops trailing zeros */
    if (X >= 0) {
        int i = 0;
        for (; i <= X; i++) *p++ = i < nd ? d[i] : '0';
        if (nd > X + 1) { *p++ = '.'; for (; i < nd; i++) *p++ = d[i]; }
    } else {
        *p++ = '0'; *p++ = '.';
        for (int z = 0; z < -X - 1; z++) *p++ = '0';
        for (int i = 0; i < nd; i++) *p++ = d[i];
    }
    return p;
}

} // namespace mtbhost
#endif
