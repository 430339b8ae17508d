/* format_check.cpp -- test harness: put_float_g6 / put_int of the host driver (metabuli_amd/csrc/host/format.h) against printf
 * over a few hundred million floats: every float of [0.999, 1], a dense sweep of bit patterns of [1e-5, 2e6], the neighbours of the powers of ten.
 * usage: format_check [stride]   prints "OK <count>" or the first differences */
#include "../../metabuli_amd/csrc/host/format.h"
#include <cstdlib>
int main(int argc, char **argv) {
    const uint32_t stride = argc > 1 ? (uint32_t)atoi(argv[1]) : 7;
    unsigned long long n = 0, bad = 0;
    auto check = [&](float v) {
        char a[64], b[64];
        char *e = mtbhost::put_float_g6(a, v); *e = 0;
        snprintf(b, sizeof(b), "%g", (double)v);
        n++;
        if (strcmp(a, b)) { if (bad < 20) printf("DIFF %a: got %s want %s\n", (double)v, a, b); bad++; }
    };
    float lo = 1e-5f, hi = 2e6f; uint32_t blo, bhi; memcpy(&blo, &lo, 4); memcpy(&bhi, &hi, 4);
    for (uint32_t b = blo; b <= bhi; b += stride) { float v; memcpy(&v, &b, 4); check(v); }
    { float a = 0.999f, z = 1.0f; uint32_t ba, bz; memcpy(&ba, &a, 4); memcpy(&bz, &z, 4); for (uint32_t b = ba; b <= bz; b++) { float v; memcpy(&v, &b, 4); check(v); } }
    for (int e = -6; e <= 7; e++) {
        float c = (float)std::pow(10.0, e);
        float v = c; for (int i = 0; i < 200; i++) { check(v); v = std::nextafterf(v, 0.0f); }
        v = c; for (int i = 0; i < 200; i++) { check(v); v = std::nextafterf(v, 1e30f); }
    }
    for (int i = 0; i <= 100000; i++) { check((float)i / 100000.0f); check((float)i / 3.0f); check((float)i * 0.5f / 150.0f); }
    check(0.0f); check(-0.0f); check(-0.5f); check(1.0f); check(INFINITY); check(NAN); check(1e-30f); check(3e38f); check(1.17549435e-38f / 4);
    char t[32];
    for (long long v : {0ll, 1ll, -1ll, 9ll, 10ll, 99ll, 1234567890123ll, -9223372036854775807ll - 1, 9223372036854775807ll}) {
        char *e = mtbhost::put_int(t, v); *e = 0; char w[32]; snprintf(w, sizeof(w), "%lld", v);
        n++; if (strcmp(t, w)) { printf("DIFF int %lld: got %s\n", v, t); bad++; }
    }
    if (bad) { printf("FAILED %llu of %llu\n", bad, n); return 1; }
    printf("OK %llu\n", n);
    return 0;
}
