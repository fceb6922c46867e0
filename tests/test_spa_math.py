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
        printf("n=%ld bad=%ld\n", n, bad);
        return bad != 0;
    }
''')


def test_spa_math_matches_host_libm(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text(SRC)
    exe = tmp_path / "t"
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-I", os.path.join(ROOT, "mercury_amd", "csrc"),
                    "-o", str(exe), str(src)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    assert "bad=0" in r.stdout
