"""CPU test: mercury_amd/csrc/glibc_trig.h (the atan / sincos the GPU front-end evaluates in restore_channel_amplitude and the
receive mixer) compiled for the host must agree BIT FOR BIT with the host libm the reference calls (misc.cc:34-71,
ofdm.cc:2331-2332). The atan restated is glibc's FMA multiarch build, so the comparison is made where the host CPU selects that
build (FMA + AVX2: every x86-64 CPU of the last decade); -fno-builtin keeps the compiler from folding libm calls itself."""
import os
import subprocess
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = textwrap.dedent(r'''
    #include <cmath>
    #include <cstdio>
    #include <cstdint>
    #include <cstring>
    #include <random>
    #include "glibc_trig.h"
    static uint64_t bits(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
    int main() {
        std::mt19937_64 rng(7);
        std::uniform_real_distribution<double> U(0, 1);
        long bad = 0, n = 0;
        auto A = [&](double x) { if (bits(atan(x)) != bits(gl_atan(x))) { if (bad < 5) printf("atan %a\n", x); ++bad; } ++n; };
        auto S = [&](double x) { double s, c, s2, c2; sincos(x, &s, &c); gl_sincos(x, &s2, &c2);
                                 if (bits(s) != bits(s2) || bits(c) != bits(c2)) { if (bad < 5) printf("sincos %a\n", x); ++bad; } ++n; };
        for (long i = 0; i < 4000000; ++i) {
            const double sg = (i & 1) ? -1.0 : 1.0;
            A(sg * std::exp((U(rng) * 120 - 60) * 0.6931471805599453)); A(sg * U(rng)); A(sg * (1 + 15 * U(rng))); A(sg * U(rng) / 16); A(sg * (16 + U(rng) * 1000));
            S(sg * U(rng) * 3.15); S(sg * U(rng) * 0.9); S(sg * std::exp((U(rng) * 40 - 38) * 0.6931471805599453)); S(sg * (0.8 + U(rng) * 1.7)); S(sg * U(rng) * 1e8);
        }
        // range boundaries of both routines, a few thousand neighbours each side
        const double ae[] = {0x1.bb67ap-27, 0.0625, 1.0, 16.0, 0x1.49ff2p+52, 0.0, 1e-310, 1e300};
        for (double e : ae) for (int v = -3000; v <= 3000; ++v) { const double x = e * (1.0 + v * 0x1p-52); A(x); A(-x); }
        for (int i = 16; i <= 256; ++i) for (int v = -200; v <= 200; ++v) { const double x = i / 256.0 * (1.0 + v * 0x1p-52); A(x); A(-x); A(1 / x); }   // table cell edges
        const double se[] = {0x1p-27, 0.126, 0.855469, 2.426265, 3.141592653589793, 1.5707963267948966, 105414350.0 * 0.999999, 0.0};
        for (double e : se) for (int v = -3000; v <= 3000; ++v) { const double x = e * (1.0 + v * 0x1p-52); S(x); S(-x); }
        for (int k = 0; k < 440; ++k) for (int v = -100; v <= 100; ++v) { const double x = (k + 0.5) / 128.0 * (1.0 + v * 0x1p-52); S(x); S(-x); }       // table cell edges
        printf("n=%ld bad=%ld\n", n, bad);
        return bad != 0;
    }
''')


def _host_has_fma():
    try:
        flags = open("/proc/cpuinfo").read()
    except OSError:
        return False
    return " fma " in flags and " avx2 " in flags


@pytest.mark.skipif(not _host_has_fma(), reason="the host libm selects a non-FMA atan on this CPU")
def test_glibc_trig_matches_host_libm(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text(SRC)
    exe = tmp_path / "t"
    subprocess.run(["g++", "-O2", "-fno-builtin", "-ffp-contract=off", "-mfma", "-I", os.path.join(ROOT, "mercury_amd", "csrc"),
                    "-o", str(exe), str(src)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    assert "bad=0" in r.stdout


def test_tables_header_matches_the_local_libm():
    """glibc_trig_tables.h is derived data: regenerate it from this machine's libm (when it is the glibc the tables came from)
    and compare."""
    libm = "/lib/x86_64-linux-gnu/libm.so.6"
    if not os.path.exists(libm):
        pytest.skip("no glibc libm at the usual place")
    r = subprocess.run(["python3", os.path.join(ROOT, "mercury_amd", "data", "gen_glibc_trig_tables.py"), libm], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("tables not found in this libm: " + r.stderr.strip())
    assert r.stdout == open(os.path.join(ROOT, "mercury_amd", "csrc", "glibc_trig_tables.h")).read()
