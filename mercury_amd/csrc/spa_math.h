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
// (compiled for the host) against the host libm on millions of arguments, including every branch.
//
// Build note: the including TU must be compiled with -ffp-contract=off.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define SPA_FN __device__ __forceinline__
#define SPA_BITS_HI(x) uint32_t(__double2hiint(x))
#define SPA_MAKE(hi, lo) __hiloint2double(int(hi), int(lo))
#define SPA_LO(x) uint32_t(__double2loint(x))
#else
#include <string.h>
#define SPA_FN static inline
static inline uint32_t spa_bits_hi_(double x) { uint64_t u; memcpy(&u, &x, 8); return uint32_t(u >> 32); }
static inline uint32_t spa_bits_lo_(double x) { uint64_t u; memcpy(&u, &x, 8); return uint32_t(u); }
static inline double spa_make_(uint32_t hi, uint32_t lo) { uint64_t u = (uint64_t(hi) << 32) | lo; double x; memcpy(&x, &u, 8); return x; }
#define SPA_BITS_HI(x) spa_bits_hi_(x)
#define SPA_LO(x) spa_bits_lo_(x)
#define SPA_MAKE(hi, lo) spa_make_(hi, lo)
#endif

SPA_FN double spa_set_high(double x, uint32_t hi) { return SPA_MAKE(hi, SPA_LO(x)); }
SPA_FN double spa_fabs(double x) { return SPA_MAKE(SPA_BITS_HI(x) & 0x7fffffffu, SPA_LO(x)); }

// expm1 for finite |x| (the decoder never feeds it inf/nan/overflowing arguments: |x| <= 44)
SPA_FN double spa_expm1(double x) {
    const double one = 1.0, tiny = 1.0e-300, huge = 1.0e+300;
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 invln2 = 1.44269504088896338700e+00;
    const double Q1 = -3.33333333333331316428e-02, Q2 = 1.58730158725481460165e-03,
                 Q3 = -7.93650757867487942473e-05, Q4 = 4.00821782732936239552e-06,
                 Q5 = -2.01099218183624371326e-07;
    double y, hi, lo, c = 0, t, e, hxs, hfx, r1, h2, h4, R1, R2, R3;
    int32_t k;
    uint32_t hx = SPA_BITS_HI(x);
    const uint32_t xsb = hx & 0x80000000u;
    hx &= 0x7fffffffu;
    if (hx >= 0x4043687Au) {            // |x| >= 56 ln2
        if (xsb != 0) return tiny - one;  // -1
        // large positive arguments fall through to the general path (k > 56)
    }
    if (hx > 0x3fd62e42u) {             // |x| > 0.5 ln2
        if (hx < 0x3FF0A2B2u) {         // |x| < 1.5 ln2
            if (xsb == 0) { hi = x - ln2_hi; lo = ln2_lo; k = 1; }
            else { hi = x + ln2_hi; lo = -ln2_lo; k = -1; }
        } else {
            k = int32_t(invln2 * x + ((xsb == 0) ? 0.5 : -0.5));
            t = k;
            hi = x - t * ln2_hi;
            lo = t * ln2_lo;
        }
        x = hi - lo;
        c = (hi - x) - lo;
    } else if (hx < 0x3c900000u) {      // |x| < 2^-54
        t = huge + x;
        return x - (t - (huge + x));
    } else {
        k = 0;
    }
    hfx = 0.5 * x;
    hxs = x * hfx;
    R1 = one + hxs * Q1; h2 = hxs * hxs;
    R2 = Q2 + hxs * Q3; h4 = h2 * h2;
    R3 = Q4 + hxs * Q5;
    r1 = R1 + h2 * R2 + h4 * R3;
    t = 3.0 - r1 * hfx;
    e = hxs * ((r1 - t) / (6.0 - x * t));
    if (k == 0) return x - (x * e - hxs);
    e = (x * (e - c) - c);
    e -= hxs;
    if (k == -1) return 0.5 * (x - e) - 0.5;
    if (k == 1) {
        if (x < -0.25) return -2.0 * (e - (x + 0.5));
        return one + 2.0 * (x - e);
    }
    if (k <= -2 || k > 56) {
        y = one - (e - x);
        y = spa_set_high(y, SPA_BITS_HI(y) + (uint32_t(k) << 20));
        return y - one;
    }
    t = one;
    if (k < 20) {
        t = spa_set_high(t, 0x3ff00000u - (0x200000u >> k));   // 1 - 2^-k
        y = t - (e - x);
        y = spa_set_high(y, SPA_BITS_HI(y) + (uint32_t(k) << 20));
    } else {
        t = spa_set_high(t, uint32_t(0x3ff - k) << 20);        // 2^-k
        y = x - (e + t);
        y += one;
        y = spa_set_high(y, SPA_BITS_HI(y) + (uint32_t(k) << 20));
    }
    return y;
}

SPA_FN double spa_tanh(double x) {
    const double one = 1.0, two = 2.0;
    double t, z;
    const uint32_t jx = SPA_BITS_HI(x), lx = SPA_LO(x);
    const uint32_t ix = jx & 0x7fffffffu;
    if (ix >= 0x7ff00000u) return (jx >> 31) ? one / x - one : one / x + one;
    if (ix < 0x40360000u) {             // |x| < 22
        if ((ix | lx) == 0) return x;
        if (ix < 0x3c800000u) return x * (one + x);   // |x| < 2^-55
        if (ix >= 0x3ff00000u) {        // |x| >= 1
            t = spa_expm1(two * spa_fabs(x));
            z = one - two / (t + two);
        } else {
            t = spa_expm1(-two * spa_fabs(x));
            z = -t / (t + two);
        }
    } else {
        z = one;                        // 1 - tiny rounds to 1
    }
    return (jx >> 31) ? -z : z;
}

// log1p for -1 < x < +inf, finite
SPA_FN double spa_log1p(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lp1 = 6.666666666666735130e-01, Lp2 = 3.999999999940941908e-01, Lp3 = 2.857142874366239149e-01,
                 Lp4 = 2.222219843214978396e-01, Lp5 = 1.818357216161805012e-01, Lp6 = 1.531383769920937332e-01,
                 Lp7 = 1.479819860511658591e-01;
    double hfsq, f = 0, c = 0, s, z, R, u, z2, z4, z6, R1, R2, R3, R4;
    int32_t k, hx, hu = 0, ax;
    hx = int32_t(SPA_BITS_HI(x));
    ax = hx & 0x7fffffff;
    k = 1;
    if (hx < 0x3FDA827A) {              // x < 0.41422
        if (ax < 0x3e200000) {          // |x| < 2^-29
            if (ax < 0x3c900000) return x;
            return x - x * x * 0.5;
        }
        if (hx > 0 || hx <= int32_t(0xbfd2bec3u)) { k = 0; f = x; hu = 1; }   // -0.2929 < x < 0.41422
    }
    if (k != 0) {
        if (hx < 0x43400000) {
            u = 1.0 + x;
            hu = int32_t(SPA_BITS_HI(u));
            k = (hu >> 20) - 1023;
            c = (k > 0) ? 1.0 - (u - x) : x - (u - 1.0);
            c /= u;
        } else {
            u = x;
            hu = int32_t(SPA_BITS_HI(u));
            k = (hu >> 20) - 1023;
            c = 0;
        }
        hu &= 0x000fffff;
        if (hu < 0x6a09e) {
            u = spa_set_high(u, uint32_t(hu) | 0x3ff00000u);
        } else {
            k += 1;
            u = spa_set_high(u, uint32_t(hu) | 0x3fe00000u);
            hu = (0x00100000 - hu) >> 2;
        }
        f = u - 1.0;
    }
    hfsq = 0.5 * f * f;
    if (hu == 0) {                      // |f| < 2^-20
        if (f == 0.0) {
            if (k == 0) return 0.0;
            c += k * ln2_lo;
            return k * ln2_hi + c;
        }
        R = hfsq * (1.0 - 0.66666666666666666 * f);
        if (k == 0) return f - R;
        return k * ln2_hi - ((R - (k * ln2_lo + c)) - f);
    }
    s = f / (2.0 + f);
    z = s * s;
    R1 = z * Lp1; z2 = z * z;
    R2 = Lp2 + z * Lp3; z4 = z2 * z2;
    R3 = Lp4 + z * Lp5; z6 = z4 * z2;
    R4 = Lp6 + z * Lp7;
    R = R1 + z2 * R2 + z4 * R3 + z6 * R4;
    if (k == 0) return f - (hfsq - s * (hfsq + R));
    return k * ln2_hi - ((hfsq - (s * (hfsq + R) + (k * ln2_lo + c))) - f);
}

// atanh for |x| < 1 (the decoder clamps +-1 to +-0.9999999 first)
SPA_FN double spa_atanh(double x) {
    const double xa = spa_fabs(x);
    double t;
    if (xa < 0.5) {
        if (xa < 0x1.0p-28) return x;
        t = xa + xa;
        t = 0.5 * spa_log1p(t + t * xa / (1.0 - xa));
    } else {
        t = 0.5 * spa_log1p((xa + xa) / (1.0 - xa));
    }
    return (SPA_BITS_HI(x) >> 31) ? -t : t;
}
