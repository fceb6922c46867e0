// tanh / atanh for the sum-product decoder, evaluated with the SAME algorithm the reference's
// libm uses, so that the GPU decoder reproduces the CPU decoder's messages bit for bit.
//
// The reference calls tanh() and atanh() from libm (ldpc_decoder_SPA.cc:145,156). On the
// reference platform (x86-64 glibc 2.35) those are the classic Sun fdlibm routines:
//     tanh(x)  = f(expm1(+-2|x|))            (s_tanh.c)
//     atanh(x) = 0.5*log1p(2x/(1-x)) forms   (e_atanh.c)
//     expm1, log1p                            (s_expm1.c, s_log1p.c: argument reduction by k*ln2,
//                                             a degree-5 / degree-7 minimax polynomial)
// They are plain IEEE-754 double arithmetic with no FMA, so restating the published algorithm
// (Sun Microsystems 1993, "Developed at SunPro ... Permission to use, copy, modify, and distribute
// this software is freely granted, provided that this notice is preserved.") with FMA contraction
// disabled gives identical bits on any IEEE machine. tests/test_spa_math.py checks this header
// (compiled for the host) against the host libm on millions of arguments, including every branch,
// and tests/test_gpu_parity.py::test_spa_math_on_device does the same for the device build.
//
// GPU shaping: the textbook routines are a tree of data-dependent branches; inside a 64-lane
// wavefront neighbouring edges take different branches almost always, so every branch body would
// be executed. The functions below perform, per lane, exactly the floating-point operations of
// the branch that lane's argument selects, but share everything the branches have in common
// (one expm1 body per tanh, one division per tanh beyond it, one log1p body per atanh) and pick
// operands/results with selects. Each select only chooses between values the original would have
// computed with the same operations, so results stay bit-identical. Branches the decoder's
// argument ranges cannot reach are dropped (stated at each function).
//
// Division: IEEE-correct a/b on gfx950 is v_div_scale x2 + v_rcp + Newton/residual FMAs +
// v_div_fmas + v_div_fixup. Scaling and fix-up only act on denormal/huge/non-finite operands; every
// division below has operands in a stated normal range, so spa_div keeps the correctly-rounding core
// (rcp, two Newton steps, quotient, residual, final FMA) and drops the three guard instructions.
//
// Build note: the including TU must be compiled with -ffp-contract=off.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define SPA_FN __device__ __forceinline__
#define SPA_BITS_HI(x) uint32_t(__double2hiint(x))
#define SPA_MAKE(hi, lo) __hiloint2double(int(hi), int(lo))
#define SPA_LO(x) uint32_t(__double2loint(x))
SPA_FN double spa_div(double n, double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(e, r, r);
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(e, r, r);
    const double q = n * r;
    const double rem = __builtin_fma(-d, q, n);
    return __builtin_fma(rem, r, q);
}
#else
#include <string.h>
#define SPA_FN static inline
static inline uint32_t spa_bits_hi_(double x) { uint64_t u; memcpy(&u, &x, 8); return uint32_t(u >> 32); }
static inline uint32_t spa_bits_lo_(double x) { uint64_t u; memcpy(&u, &x, 8); return uint32_t(u); }
static inline double spa_make_(uint32_t hi, uint32_t lo) { uint64_t u = (uint64_t(hi) << 32) | lo; double x; memcpy(&x, &u, 8); return x; }
#define SPA_BITS_HI(x) spa_bits_hi_(x)
#define SPA_LO(x) spa_bits_lo_(x)
#define SPA_MAKE(hi, lo) spa_make_(hi, lo)
SPA_FN double spa_div(double n, double d) { return n / d; }
#endif

SPA_FN double spa_set_high(double x, uint32_t hi) { return SPA_MAKE(hi, SPA_LO(x)); }
SPA_FN double spa_fabs(double x) { return SPA_MAKE(SPA_BITS_HI(x) & 0x7fffffffu, SPA_LO(x)); }
SPA_FN double spa_add_exponent(double y, int32_t k) { return spa_set_high(y, SPA_BITS_HI(y) + (uint32_t(k) << 20)); }

// tanh(x) for any finite x (s_tanh.c + s_expm1.c).
// expm1 is only ever evaluated at -2|x| in [-2, -2^-54] (|x| < 1) or at 2|x| in [2, 44) (1 <= |x| < 22),
// so of s_expm1.c's cases k = 0, k = -1, k <= -2, 2 <= k < 20, 20 <= k <= 56 and k > 56 remain
// (k = +1, the |x| < 2^-54 shortcut and the x <= -56 ln2 shortcut cannot occur).
SPA_FN double spa_tanh(double x) {
    const double one = 1.0, two = 2.0;
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 invln2 = 1.44269504088896338700e+00;
    const double Q1 = -3.33333333333331316428e-02, Q2 = 1.58730158725481460165e-03,
                 Q3 = -7.93650757867487942473e-05, Q4 = 4.00821782732936239552e-06,
                 Q5 = -2.01099218183624371326e-07;
    const uint32_t jx = SPA_BITS_HI(x);
    const uint32_t ix = jx & 0x7fffffffu;
    const double ax = spa_fabs(x);
    const bool big = ix >= 0x3ff00000u;                      // |x| >= 1
    // ---- t = expm1(big ? 2|x| : -2|x|) -------------------------------------------------------
    double a = two * ax;                                     // exact
    a = big ? a : -a;
    const uint32_t ha = SPA_BITS_HI(a) & 0x7fffffffu;
    const bool kzero = !(ha > 0x3fd62e42u);                  // |a| <= 0.5 ln2
    const bool kone = !kzero && (ha < 0x3FF0A2B2u);          // 0.5 ln2 < |a| < 1.5 ln2 (only reached with a < 0)
    int32_t k = int32_t(invln2 * a + (big ? 0.5 : -0.5));
    k = kone ? -1 : k;
    k = kzero ? 0 : k;
    const double tk = double(k);
    const double hi = a - tk * ln2_hi;                       // exact products for k = 0, -1
    const double lo = tk * ln2_lo;
    const double r = hi - lo;
    const double c = (hi - r) - lo;                          // 0 when k == 0
    const double hfx = 0.5 * r;
    const double hxs = r * hfx;
    const double R1 = one + hxs * Q1, h2 = hxs * hxs;
    const double R2 = Q2 + hxs * Q3, h4 = h2 * h2;
    const double R3 = Q4 + hxs * Q5;
    const double r1 = R1 + h2 * R2 + h4 * R3;
    const double tt = 3.0 - r1 * hfx;
    double e = hxs * spa_div(r1 - tt, 6.0 - r * tt);         // denominator in [5, 7]
    const double res0 = r - (r * e - hxs);                   // k == 0
    e = (r * (e - c) - c);
    e -= hxs;
    const double resm1 = 0.5 * (r - e) - 0.5;                // k == -1
    const double emx = e - r;
    const double ya = spa_add_exponent(one - emx, k) - one;                              // k <= -2 || k > 56
    const double tb = SPA_MAKE(0x3ff00000u - (0x200000u >> (uint32_t(k) & 31u)), 0u);   // 1 - 2^-k, 2 <= k < 20
    const double yb = spa_add_exponent(tb - emx, k);
    double t = (k <= -2) ? ya : yb;
    t = (k == -1) ? resm1 : t;
    t = (k == 0) ? res0 : t;
    if (k >= 20) {                       // |x| > 6.7: rare at the SNRs where the decoder iterates; a real branch
        const double tc = SPA_MAKE(uint32_t(0x3ff - k) << 20, 0u);                       // 2^-k, 20 <= k <= 56
        const double yc = spa_add_exponent((r - (e + tc)) + one, k);
        t = (k > 56) ? ya : yc;
    }
    // ---- tanh from t -------------------------------------------------------------------------
    const double q = spa_div(big ? two : -t, t + two);       // denominator in [1, 2^64]
    double z = big ? one - q : q;
    z = (ix >= 0x40360000u) ? one : z;                       // |x| >= 22: one - tiny
    const double res = (jx >> 31) ? -z : z;
    // |x| < 2^-55 (incl. +-0): x*(one+x) == x ; non-finite arguments never reach the decoder
    return (ix < 0x3c800000u) ? x : res;
}

// atanh(x) for |x| < 1 (e_atanh.c + s_log1p.c; the decoder clamps +-1 to +-0.9999999 first).
// log1p is only ever evaluated at y = 2|x|/(1-|x|)-type arguments with 2^-27 <= y <= 2e7, so of
// s_log1p.c's cases the y <= -0.2929, |y| < 2^-29 and y >= 2^53 ones cannot occur.
SPA_FN double spa_atanh(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lp1 = 6.666666666666735130e-01, Lp2 = 3.999999999940941908e-01, Lp3 = 2.857142874366239149e-01,
                 Lp4 = 2.222219843214978396e-01, Lp5 = 1.818357216161805012e-01, Lp6 = 1.531383769920937332e-01,
                 Lp7 = 1.479819860511658591e-01;
    const double xa = spa_fabs(x);
    const bool small = xa < 0.5;
    const double t2 = xa + xa;
    const double q = spa_div(small ? t2 * xa : t2, 1.0 - xa);    // denominator in [1e-7, 1]
    const double y = small ? t2 + q : q;
    // ---- log1p(y), y > 0 ----------------------------------------------------------------------
    const int32_t hy = int32_t(SPA_BITS_HI(y));
    const bool direct = hy < 0x3FDA827A;                     // y < 0.41422: f = y, k = 0
    double u = 1.0 + y;
    int32_t hu = int32_t(SPA_BITS_HI(u));
    int32_t k = (hu >> 20) - 1023;
    double c = (k > 0) ? 1.0 - (u - y) : y - (u - 1.0);
    c = spa_div(c, u);                                       // u in [1, 2e7]; |c| <= 2^-53 or 0
    hu &= 0x000fffff;
    const bool lowhalf = hu < 0x6a09e;
    u = spa_set_high(u, uint32_t(hu) | (lowhalf ? 0x3ff00000u : 0x3fe00000u));
    k = lowhalf ? k : k + 1;
    hu = lowhalf ? hu : (0x00100000 - hu) >> 2;
    double f = u - 1.0;
    f = direct ? y : f;
    k = direct ? 0 : k;
    c = direct ? 0.0 : c;
    hu = direct ? 1 : hu;
    const double dk = double(k);
    const double hfsq = 0.5 * f * f;
    double l;
    if (hu == 0) {                       // |f| < 2^-20: rare, short
        if (f == 0.0) {
            l = (k == 0) ? 0.0 : dk * ln2_hi + (c + dk * ln2_lo);
        } else {
            const double R = hfsq * (1.0 - 0.66666666666666666 * f);
            l = (k == 0) ? f - R : dk * ln2_hi - ((R - (dk * ln2_lo + c)) - f);
        }
    } else {
        const double s = spa_div(f, 2.0 + f);                // denominator in [1.7, 2.42]
        const double z = s * s;
        const double R1 = z * Lp1, z2 = z * z;
        const double R2 = Lp2 + z * Lp3, z4 = z2 * z2;
        const double R3 = Lp4 + z * Lp5, z6 = z4 * z2;
        const double R4 = Lp6 + z * Lp7;
        const double R = R1 + z2 * R2 + z4 * R3 + z6 * R4;
        const double sr = s * (hfsq + R);
        l = (k == 0) ? f - (hfsq - sr) : dk * ln2_hi - ((hfsq - (sr + (dk * ln2_lo + c))) - f);
    }
    double t = 0.5 * l;
    t = (SPA_BITS_HI(x) >> 31) ? -t : t;
    return (xa < 0x1.0p-28) ? x : t;
}
