"""CPU test: mercury_amd/csrc/spa_math.h (the tanh/atanh the GPU sum-product decoder evaluates) compiled
for the host must agree BIT FOR BIT with the host libm the reference calls (ldpc_decoder_SPA.cc:145,156)."""
import os
import subprocess
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = textwrap.dedent(r'''
    #include <cmath>
    #include <cstdio>
    #include <cstdint>
    #include <cstring>
    #include <random>
    #include "spa_math.h"
    static uint64_t bits(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
    int main() {
        std::mt19937_64 rng(2024);
        std::uniform_real_distribution<double> U(0, 1);
        long bad = 0, n = 0;
        auto T = [&](double x) { if (bits(tanh(x)) != bits(spa_tanh(x))) { if (bad < 5) printf("tanh %a\n", x); ++bad; } ++n; };
        auto A = [&](double x) { if (bits(atanh(x)) != bits(spa_atanh(x))) { if (bad < 5) printf("atanh %a\n", x); ++bad; } ++n; };
        for (long i = 0; i < 3000000; ++i) {
            double u = U(rng), s = (i & 1) ? -1.0 : 1.0;
            T(s * std::exp((U(rng) * 62 - 46) * 0.6931471805599453));   // 2^-46 .. 2^16
            T((u * 2 - 1) * 25);
            A(u * 2 - 1);
            double y = std::tanh((u * 2 - 1) * 20); if (std::fabs(y) < 1) A(y);
            y = s * (1 - std::exp(-U(rng) * 36)); if (std::fabs(y) < 1) A(y);
            A(s * std::exp(-U(rng) * 40));
        }
        const double edge[] = {0.0, -0.0, 1.0, -1.0, 0.5, 21.999999, 22.0, 23.0, 1e-300, 0x1p-55, 0x1p-54, 0.34657359027997264,
                               1.0397207708399179, 19.061547465398498, 38.0, 44.0, 709.0};
        for (double x : edge) { T(x); T(-x); }
        const double aedge[] = {0.9999999, -0.9999999, 0.5, 0x1p-28, 0x1p-29, 0.0, 1 - 0x1p-53, 0.41422, 0.2929};
        for (double x : aedge) { A(x); A(-x); }
        // the decoder's call forms: tanh(0.5*q) and 2*atanh(clamp(x))  (ldpc_decoder_SPA.cc:145-156)
        auto TH = [&](double q) { if (bits(tanh(0.5 * q)) != bits(spa_tanh_half(q))) { if (bad < 5) printf("tanh_half %a\n", q); ++bad; } ++n; };
        auto A2 = [&](double x) {
            double c = x; if (c == 1) c = 0.9999999; if (c == -1) c = -0.9999999;
            if (bits(2 * atanh(c)) != bits(spa_atanh_x2(x))) { if (bad < 5) printf("atanh_x2 %a\n", x); ++bad; } ++n; };
        for (long i = 0; i < 3000000; ++i) {
            double u = U(rng), s = (i & 1) ? -1.0 : 1.0;
            TH(s * std::exp((U(rng) * 62 - 46) * 0.6931471805599453));
            TH((u * 2 - 1) * 50);
            TH((u * 2 - 1) * 3);
            A2(u * 2 - 1);
            double y = s * (1 - std::exp(-U(rng) * 36)); A2(y);
            A2(s * std::exp(-U(rng) * 40));
            A2(std::tanh((u * 2 - 1) * 3) * std::tanh(U(rng) * 3) * std::tanh(U(rng) * 3));
        }
        A2(1.0); A2(-1.0);
        // the decoder's entry points with their wavefront-wide shortcuts (on the host a wavefront is one lane)
        auto THW = [&](double q) { if (bits(tanh(0.5 * q)) != bits(spa_tanh_half_wave(q))) { if (bad < 5) printf("tanh_half_wave %a\n", q); ++bad; } ++n; };
        auto A2W = [&](double x) {
            double c = x; if (c == 1) c = 0.9999999; if (c == -1) c = -0.9999999;
            if (bits(2 * atanh(c)) != bits(spa_atanh_x2_wave(x))) { if (bad < 5) printf("atanh_x2_wave %a\n", x); ++bad; } ++n; };
        for (long i = 0; i < 200000; ++i) {
            double u = U(rng), s = (i & 1) ? -1.0 : 1.0;
            THW(s * std::exp((U(rng) * 62 - 46) * 0.6931471805599453)); THW((u * 2 - 1) * 100); THW(s * (44.0 + (u - 0.5) * 1e-9));
            A2W(u * 2 - 1); A2W(s * std::exp(-U(rng) * 40)); A2W(s * 0x1p-28 * (1 + (u - 0.5) * 1e-9));
        }
        for (double q : {44.0, 43.99999999999999, 44.00000000000001, 1e300, 1e-300}) { THW(q); THW(-q); }
        THW(INFINITY); THW(-INFINITY);
        A2W(1.0); A2W(-1.0); A2W(0.0); A2W(-0.0); A2W(0x1p-28); A2W(-0x1p-28); A2W(0x1p-29);
        if (bits(spa_tanh_half_wave(NAN)) != bits(spa_tanh_half(NAN))) { printf("tanh_half_wave NaN\n"); ++bad; }
        // every high-word threshold of the two routines, swept through the words around it with extreme and random low words
        const uint32_t th_q[] = {0x3c900000u, 0x3c800000u, 0x3fd62e42u, 0x3fd62e43u, 0x3ff0a2b2u, 0x3ff00000u, 0x40000000u, 0x40038000u,
                                 0x402b0000u, 0x402bb9d3u, 0x402bb9d4u, 0x40434e00u, 0x40436800u, 0x40460000u, 0x40450000u, 0x3fe62e42u};
        for (uint32_t h : th_q)
            for (int dh = -3; dh <= 3; ++dh)
                for (int v = 0; v < 4096; ++v) {
                    const uint32_t lo = v == 0 ? 0u : v == 1 ? 1u : v == 2 ? 0xffffffffu : v == 3 ? 0xfffffffeu : uint32_t(rng());
                    uint64_t b = (uint64_t(h + dh) << 32) | lo; double q; memcpy(&q, &b, 8);
                    TH(q); TH(-q);
                }
        // k boundaries of expm1: |q| = (k + 0.5) * ln2 for every k the decoder can reach, +- a few ulps and random nearby
        for (int k = 0; k < 64; ++k) {
            const double c0 = (k + 0.5) * 0.6931471805599453;
            for (int v = -2000; v <= 2000; ++v) { const double q = c0 * (1.0 + v * 0x1p-52); TH(q); TH(-q); }
            for (int v = 0; v < 2000; ++v) { const double q = c0 * (1.0 + (U(rng) - 0.5) * 1e-6); TH(q); TH(-q); }
        }
        const uint32_t th_a[] = {0x3fe00000u, 0x3e300000u, 0x3fc5f619u, 0x3fc5f61au, 0x3fd00000u, 0x3fd55555u, 0x3fefffffu, 0x3feffffeu};
        for (uint32_t h : th_a)
            for (int dh = -3; dh <= 3; ++dh)
                for (int v = 0; v < 4096; ++v) {
                    const uint32_t lo = v == 0 ? 0u : v == 1 ? 1u : v == 2 ? 0xffffffffu : v == 3 ? 0xfffffffeu : uint32_t(rng());
                    uint64_t b = (uint64_t(h + dh) << 32) | lo; double x; memcpy(&x, &b, 8);
                    if (std::fabs(x) <= 1) { A2(x); A2(-x); }
                }
        // log1p's direct / normalised switch (y = 2x/(1-x) around 0.41422) and the u ~ sqrt(2) * 2^k normalisation switches
        for (int k = 0; k < 26; ++k) {
            const double u0 = std::ldexp(1.4142131805419922, k), y0 = u0 - 1, x0 = y0 / (2 + y0);
            for (int v = 0; v < 20000; ++v) { const double x = x0 * (1.0 + (U(rng) - 0.5) * 4e-7); if (x < 1) { A2(x); A2(-x); } }
            const double y1 = std::ldexp(1.0, k + 1) - 1, x1 = y1 / (2 + y1);
            for (int v = 0; v < 20000; ++v) { const double x = x1 * (1.0 + (U(rng) - 0.5) * 4e-7); if (x < 1) { A2(x); A2(-x); } }
        }
        // tiny arguments: every binade from the smallest denormal up to 2^-20 (tanh has no branch of its own for them; atanh's is |x| < 2^-28)
        for (int e = -1074; e <= -20; ++e)
            for (int v = 0; v < 64; ++v) {
                const double m = v == 0 ? 1.0 : v == 1 ? 2.0 - 0x1p-52 : 1.0 + U(rng);
                const double x = std::ldexp(m, e);
                T(x); T(-x); TH(x); TH(-x); A2(x); A2(-x); A(x); A(-x);
            }
        for (uint64_t b = 0; b < 4096; ++b) { double x; memcpy(&x, &b, 8); TH(x); TH(-x); T(x); A2(x); A2(-x); }   // the smallest denormals
        printf("n=%ld bad=%ld\n", n, bad);
        return bad != 0;
    }
''')


def test_spa_math_matches_host_libm(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text(SRC)
    exe = tmp_path / "t"
    subprocess.run(["g++", "-O2", "-fno-builtin", "-ffp-contract=off", "-I", os.path.join(ROOT, "mercury_amd", "csrc"),
                    "-o", str(exe), str(src)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    assert "bad=0" in r.stdout
