// mgpu_host_libm_selfcheck: is the host's libm the one the device code restates?
//
// The fp64 decoder's tanh / atanh (spa_math.h) and the front-end's atan / sincos (glibc_trig.h) restate x86-64 glibc 2.35 - the libm the
// reference called where its outputs were pinned (ldpc_decoder_SPA.cc:145,156; misc.cc:34-71; ofdm.cc:2331-2332). "Bit-identical to the
// CPU reference" therefore means: to a reference linked against a libm that computes those four functions as glibc 2.35 does. A host with
// another libm (a later glibc with new multiarch variants, musl, a vendor libm) runs the reference with ITS results, and the library
// cannot know that from the device side - so it checks on the host: the same headers, compiled here for the host, against the libm this
// process is linked to, on the arguments tests/test_spa_math.py and tests/test_glibc_trig.py concentrate on (every branch threshold of
// the routines swept through the neighbouring words, the table-cell edges, the reduction boundaries) plus a deterministic random sample.
// Host-only, no GPU needed, 60-90 ms (about a million libm calls). mgpu_create runs it only when MERCURY_GPU_LIBM_CHECK=1 is set.
#include <gnu/libc-version.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "../../include/mercury_gpu.h"
#include "glibc_trig.h"
#include "spa_math.h"

namespace {

uint64_t bits(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
double from_bits(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }

struct Lcg {                                   // deterministic arguments: the check must not depend on the libc's own rand()
    uint64_t s;
    uint64_t next() { s = s * 6364136223846793005ULL + 1442695040888963407ULL; return s; }
    double unit() { return double(next() >> 11) * 0x1p-53; }
};

struct Tally {
    long long n = 0, bad = 0;
    double first = 0;
    void note(bool same, double x) { ++n; if (!same) { if (!bad) first = x; ++bad; } }
};

void run(mgpu_libm_report* out) {
    Tally th, ah, at, sc;
    volatile double sink = 0;                  // keeps the libm calls from being folded
    auto TH = [&](double q) { const double ref = tanh(0.5 * q); sink = ref; th.note(bits(ref) == bits(spa_tanh_half(q)), q); };
    auto A2 = [&](double x) {                  // the decoder's call form: 2 * atanh(clamp(x)) (ldpc_decoder_SPA.cc:150-156)
        double c = x; if (c == 1) c = 0.9999999; if (c == -1) c = -0.9999999;
        const double ref = 2 * atanh(c); sink = ref;
        ah.note(bits(ref) == bits(spa_atanh_x2(x)), x);
    };
    auto AT = [&](double x) { const double ref = atan(x); sink = ref; at.note(bits(ref) == bits(gl_atan(x)), x); };
    auto SC = [&](double x) {
        double s, c, s2, c2;
        sincos(x, &s, &c); gl_sincos(x, &s2, &c2); sink = s + c;
        sc.note(bits(s) == bits(s2) && bits(c) == bits(c2), x);
    };
    Lcg g{0x4D455243ULL};
    // every high-word threshold of fdlibm's tanh / expm1 and atanh / log1p, +-3 words, extreme and random low words
    const uint32_t th_q[] = {0x3c900000u, 0x3c800000u, 0x3fd62e42u, 0x3fd62e43u, 0x3ff0a2b2u, 0x3ff00000u, 0x40000000u, 0x40038000u,
                             0x402b0000u, 0x402bb9d3u, 0x402bb9d4u, 0x40434e00u, 0x40436800u, 0x40460000u, 0x40450000u, 0x3fe62e42u};
    for (uint32_t h : th_q)
        for (int dh = -3; dh <= 3; ++dh)
            for (int v = 0; v < 64; ++v) {
                const uint32_t lo = v == 0 ? 0u : v == 1 ? 1u : v == 2 ? 0xffffffffu : v == 3 ? 0xfffffffeu : uint32_t(g.next() >> 32);
                const double q = from_bits((uint64_t(h + dh) << 32) | lo);
                TH(q); TH(-q);
            }
    for (int k = 0; k < 64; ++k) {             // expm1's k boundaries: |q| = (k + 0.5) ln 2
        const double c0 = (k + 0.5) * 0.6931471805599453;
        for (int v = -40; v <= 40; ++v) { const double q = c0 * (1.0 + v * 0x1p-52); TH(q); TH(-q); }
    }
    const uint32_t th_a[] = {0x3fe00000u, 0x3e300000u, 0x3fc5f619u, 0x3fc5f61au, 0x3fd00000u, 0x3fd55555u, 0x3fefffffu, 0x3feffffeu};
    for (uint32_t h : th_a)
        for (int dh = -3; dh <= 3; ++dh)
            for (int v = 0; v < 64; ++v) {
                const uint32_t lo = v == 0 ? 0u : v == 1 ? 1u : v == 2 ? 0xffffffffu : v == 3 ? 0xfffffffeu : uint32_t(g.next() >> 32);
                const double x = from_bits((uint64_t(h + dh) << 32) | lo);
                if (fabs(x) <= 1) { A2(x); A2(-x); }
            }
    for (int k = 0; k < 26; ++k) {             // log1p's u ~ sqrt(2) 2^k normalisation switches
        const double u0 = ldexp(1.4142131805419922, k), y0 = u0 - 1, x0 = y0 / (2 + y0);
        for (int v = 0; v < 200; ++v) { const double x = x0 * (1.0 + (g.unit() - 0.5) * 4e-7); if (x < 1) { A2(x); A2(-x); } }
    }
    const double edge[] = {0.0, 1.0, 0.5, 21.999999, 22.0, 23.0, 1e-300, 0x1p-55, 0x1p-54, 0.34657359027997264, 1.0397207708399179,
                           19.061547465398498, 38.0, 44.0, 709.0};
    for (double x : edge) { TH(2 * x); TH(-2 * x); }
    A2(1.0); A2(-1.0); A2(0.0); A2(1 - 0x1p-53);
    for (int i = 0; i < 60000; ++i) {          // what the decoder feeds them: LLR-sized arguments, products of tanh values
        const double u = g.unit(), s = (i & 1) ? -1.0 : 1.0;
        TH(s * exp((g.unit() * 62 - 46) * 0.6931471805599453));
        TH((u * 2 - 1) * 50);
        A2(u * 2 - 1);
        A2(s * (1 - exp(-g.unit() * 36)));
        A2(s * exp(-g.unit() * 40));
    }
    // atan: the five ranges' limits and the 241-row table's cell edges; sincos: the table's cell edges and the range limits
    for (int i = 16; i <= 256; ++i)
        for (int v = -8; v <= 8; ++v) { const double x = i / 256.0 * (1.0 + v * 0x1p-52); AT(x); AT(-x); AT(1 / x); }
    const double at_edge[] = {0x1p-27, 0.0625, 1.0, 16.0, 0x1p52, 0.0, 1e-300, 1e300};
    for (double x : at_edge) for (int v = -8; v <= 8; ++v) { const double y = x * (1.0 + v * 0x1p-52); AT(y); AT(-y); }
    for (int k = 0; k < 440; ++k)
        for (int v = -4; v <= 4; ++v) { const double x = (k + 0.5) / 128.0 * (1.0 + v * 0x1p-52); SC(x); SC(-x); }
    const double sc_edge[] = {0x1p-27, 0.126, 0.855469, 2.426265, 105414350.0 * 0.99, 0.0, 1.5707963267948966, 3.141592653589793};
    for (double x : sc_edge) for (int v = -8; v <= 8; ++v) { const double y = x * (1.0 + v * 0x1p-52); SC(y); SC(-y); }
    for (int i = 0; i < 60000; ++i) {
        const double u = g.unit();
        AT((u * 2 - 1) * 4); AT(exp((g.unit() * 80 - 40) * 0.6931471805599453)); AT(-1 / (u + 1e-9));
        SC((u * 2 - 1) * 3.2); SC((g.unit() * 2 - 1) * 1e4); SC((g.unit() * 2 - 1) * 1e7);
    }
    (void)sink;
    const Tally* t[4] = {&th, &ah, &at, &sc};
    out->differing = 0;
    for (int i = 0; i < 4; ++i) {
        out->evaluated[i] = t[i]->n; out->differed[i] = t[i]->bad; out->first_differing_argument[i] = t[i]->first;
        out->differing += t[i]->bad != 0;
    }
    snprintf(out->libc_version, sizeof out->libc_version, "glibc %s", gnu_get_libc_version());
}

}  // namespace

extern "C" int mgpu_host_libm_selfcheck(mgpu_libm_report* out) {
    static std::once_flag once;
    static mgpu_libm_report cached;
    std::call_once(once, [] { run(&cached); });
    if (out) *out = cached;
    return cached.differing;
}

// mgpu_create's side. OPT-IN (round 6): with MERCURY_GPU_LIBM_CHECK=1 in the environment the first mgpu_create of the process runs
// the check (~1 M libm calls, measured 60-90 ms on this image's host cores - too much to impose on every process, and a library should not write
// to stderr uninvited) and writes one line to stderr when something differs. Without it nothing runs here; a caller asks
// mgpu_host_libm_selfcheck() itself when it wants the report.
extern "C" void mgpu_internal_libm_notice() {
    static std::once_flag once;
    std::call_once(once, [] {
        const char* e = getenv("MERCURY_GPU_LIBM_CHECK");
        if (!e || e[0] != '1') return;
    mgpu_libm_report r;
        if (mgpu_host_libm_selfcheck(&r) == 0) return;
        static const char* name[4] = {"tanh", "atanh", "atan", "sincos"};
        fprintf(stderr, "[mercury_gpu] host libm (%s) differs from the one the device code restates (x86-64 glibc 2.35):", r.libc_version);
        for (int i = 0; i < 4; ++i)
            if (r.differed[i]) fprintf(stderr, " %s %lld of %lld (first at %a)", name[i], r.differed[i], r.evaluated[i], r.first_differing_argument[i]);
        fprintf(stderr, ". Results stay those of the glibc 2.35 reference; a reference built on THIS host may differ in the last bit of %s.\n",
                (r.differed[0] || r.differed[1]) ? "decoder messages (and, rarely, iteration counts)" : "the PSK modes' equalised grid / re-mixed windows");
    });
}
