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
// disabled gives identical bits on any IEEE machine (the fmas written out below are the ones whose
// product is exact, i.e. that round once like the reference's multiply-then-add: see "Round 5"). tests/test_spa_math.py checks this header
// (compiled for the host) against the host libm on millions of arguments, including every branch
// boundary, and tests/test_gpu_parity.py::test_spa_math_on_device does the same for the device build.
//
// GPU shaping (round 2; measured costs in profiles/r02_valu_cycles.json: an fp64 add/mul/fma issues
// in 4 cycles, v_rcp_f64 in 16, a 64-bit select is 2 x v_cndmask + a compare = 8..12):
//  * The decoder's call sites are tanh(0.5*Q) and 2*atanh(T): spa_tanh_half(Q) uses |Q| = 2|x| directly
//    (the doubling and halving are exact) and spa_atanh_x2 returns +-log1p(..) (0.5*l*2 == l exactly).
//  * expm1's argument is +-2|x|; everything up to the polynomial is odd in that sign, so the reduction
//    (k, hi, lo, r, c) is done once for |a| and the sign is put back with two XORs on high words.
//  * k comes from the general formula int(invln2*a +- 0.5) in both of fdlibm's explicit k = +-1 and
//    k = 0 regions; only the sliver between 0.5*ln2 and the high-word threshold 0x3fd62e42ffffffff
//    needs the k = 0 override (checked exhaustively around the thresholds by the host test).
//  * Where fdlibm's branches compute DIFFERENT things from the shared intermediate values (expm1's
//    k = -1 / <= -2 / 2..19 / 20..56 / > 56 endings, log1p's direct / normalised forms) each class is
//    a real divergent branch: a lane executes exactly its own class's operations and writes its result
//    under the execution mask, which costs scalar instructions only, instead of every lane computing
//    every ending and choosing with 64-bit selects. Classes no lane of the wavefront is in are skipped.
//    SPA_KEEP stops the compiler from converting those bodies back into selects.
//  * Division: IEEE-correct a/b on gfx950 is v_div_scale x2 + v_rcp + 2 Newton steps + quotient +
//    residual + v_div_fmas + v_div_fixup. Scaling and fix-up only act on denormal/huge/non-finite
//    operands; every division here has operands in a stated normal range, so spa_div keeps the
//    correctly-rounding core, with the two Newton steps (4 FMAs, error e^4) replaced by one cubic step
//    r(1 + e + e^2) (3 FMAs, error e^3 = 2^-66 from the >= 22-bit seed): rcp, 3 FMAs, quotient, residual,
//    final FMA. Checked against the host's division on 10^8 operand pairs per range on the device.
//
// Build note: the including TU must be compiled with -ffp-contract=off.
#pragma once
#include <stdint.h>
#ifndef SPA_CENSUS
#define SPA_CENSUS(i) do {} while (0)      /* ldpc.hip's branch census (variant builds) */
#endif

#if defined(__HIPCC__) || defined(__HIP__)
#define SPA_FN __device__ __forceinline__
#define SPA_BITS_HI(x) uint32_t(__double2hiint(x))
#define SPA_MAKE(hi, lo) __hiloint2double(int(hi), int(lo))
#define SPA_LO(x) uint32_t(__double2loint(x))
#define SPA_KEEP(x) asm volatile("" : "+v"(x))
#define SPA_UNDEF(x) asm volatile("" : "=v"(x))      /* a definition without an instruction */
#define SPA_ONE_COMPARE(b) (b) = __builtin_amdgcn_inverse_ballot_w64(__builtin_amdgcn_ballot_w64(b))   /* a lane condition used by a select AND a branch: compared once, kept as a lane mask */
#define SPA_ALL(c) (__builtin_amdgcn_ballot_w64(!(c)) == 0)      /* true for every active lane of the wavefront: a scalar branch */
#define SPA_ALL_OF(m, c) ((__builtin_amdgcn_ballot_w64(!(c)) & (m)) == 0)      /* ... for every lane of the mask m */
#define SPA_SGPR(x) asm volatile("" : "+s"(x))       /* a constant that is not an inline operand, kept in scalar registers (an fma's addend would otherwise be moved into vector registers) */
#ifndef SPA_VCONST
#define SPA_VCONST 575
#endif
/* A double constant that is not an inline operand, materialised in a VECTOR register pair right where it is used: two v_mov_b32 with literals
 * inside volatile asm statements (so that they are neither hoisted out of the bin loop, where they would occupy registers the loop does not
 * have, nor turned back into the scalar unit's s_mov_b32 pair). HI / LO are the constant's words (checked at compile time). */
#define SPA_VREG(bit, x, HI, LO) do { static_assert(__builtin_bit_cast(unsigned long long, x) == ((unsigned long long)(HI) << 32 | (LO)), "SPA_VREG: words do not spell the constant"); \
    if constexpr ((SPA_VCONST & (bit)) != 0) { uint32_t lo_, hi_; asm volatile("v_mov_b32 %0, " #LO : "=v"(lo_)); asm volatile("v_mov_b32 %0, " #HI : "=v"(hi_)); x##_v = SPA_MAKE(hi_, lo_); } } while (0)
SPA_FN double spa_recip(double d) {          // 1/d to within an ulp, d normal
    const double r = __builtin_amdgcn_rcp(d);
    const double e = __builtin_fma(-d, r, 1.0);
    const double p = __builtin_fma(e, e, e);
    return __builtin_fma(r, p, r);
}
SPA_FN double spa_div_r(double n, double d, double r) {   // correctly rounded n/d given r = spa_recip(d)
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
#define SPA_KEEP(x) (void)(x)
#define SPA_UNDEF(x) (x) = 0
#define SPA_SGPR(x) (void)(x)
#define SPA_VREG(bit, x, HI, LO) (void)(x)
#undef SPA_VCONST
#define SPA_VCONST 575
#define SPA_ALL(c) (c)
#define SPA_ALL_OF(m, c) (c)
#define SPA_ONE_COMPARE(b) (void)(b)
SPA_FN double spa_recip(double d) { return d; }
SPA_FN double spa_div_r(double n, double d, double) { return n / d; }
#endif

#if defined(__HIPCC__) || defined(__HIP__)
SPA_FN double spa_fabs(double x) { return __builtin_fabs(x); }     // a source modifier, no instruction
#else
SPA_FN double spa_fabs(double x) { return SPA_MAKE(SPA_BITS_HI(x) & 0x7fffffffu, SPA_LO(x)); }
#endif
SPA_FN double spa_div(double n, double d) { return spa_div_r(n, d, spa_recip(d)); }

// 1 - 2/d given rr = spa_recip(d). On the device the quotient sequence for the numerator 2 is 2 * (that for the numerator 1): q = 2 rr,
// rem = fma(-d, q, 2) = 2 fma(-d, rr, 1), fma(rem, rr, q) = 2 fma(rem/2, rr, rr) - doubling commutes with every rounding - and 1 - 2X is
// one fma (the product is exact): three instructions for four, the same bits.
#if defined(__HIPCC__) || defined(__HIP__)
SPA_FN double spa_one_minus_2_over(double d, double rr) {
    const double x = __builtin_fma(__builtin_fma(-d, rr, 1.0), rr, rr);
    return __builtin_fma(-2.0, x, 1.0);
}
#else
SPA_FN double spa_one_minus_2_over(double d, double) { return 1.0 - 2.0 / d; }
#endif

// Round 5: both routines carry some of fdlibm's intermediate values SCALED BY A POWER OF TWO (r*r = 2 hxs instead of hxs, y/2, f/2, s/2, z/4 ...)
// and put the scale back inside an fma whose product is exact (2 * a, 0.5 * a, k * ln2_hi with its 32-bit mantissa), or inside the
// polynomial's constants. Scaling by a power of two commutes with every IEEE rounding away from the denormal range - fl(2a * b) = 2 fl(a * b),
// fl(2a + 2b) = 2 fl(a + b) - so every value below is fdlibm's own, times the stated power of two, and every result has fdlibm's bits; what
// goes is the instructions that only doubled or halved (hfx = 0.5 r, t2 = x + x, 0.5 f) and the separate multiplications in front of
// additions whose product is exact. (Where an intermediate can be denormal - arguments below 2^-500 or so - the host test checks the results
// binade by binade down to the smallest denormal: atanh answers those from its |x| < 2^-28 branch, tanh's general path yields 0.5*q there.)
// tests/test_spa_math.py runs this very code on the host against the host's libm.

// tanh(0.5 * q)
SPA_FN double spa_tanh_half(double q) {
    constexpr double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                     invln2 = 1.44269504088896338700e+00;
    constexpr double Q1 = -3.33333333333331316428e-02, Q2 = 1.58730158725481460165e-03,
                     Q3 = -7.93650757867487942473e-05, Q4 = 4.00821782732936239552e-06,
                     Q5 = -2.01099218183624371326e-07;
    const uint32_t jq = SPA_BITS_HI(q), iq = jq & 0x7fffffffu;
    const double A = spa_fabs(q);                            // = 2|x| (exact for |x| >= 2^-55)
    SPA_CENSUS(0);
    bool big = iq >= 0x40000000u;                            // |x| >= 1
    SPA_ONE_COMPARE(big);
    double invln2_v = invln2;
    SPA_VREG(1, invln2, 0x3ff71547, 0x652b82fe);
    int32_t kk = int32_t(invln2_v * A + 0.5);
    kk = (iq <= 0x3fd62e42u) ? 0 : kk;
    const double tk = double(kk);
    const double hi = __builtin_fma(-tk, ln2_hi, A);         // A - tk*ln2_hi: the product is exact (|k| < 2^10, ln2_hi has 32 mantissa bits)
    const double lo = tk * ln2_lo;
    double rp = hi - lo;
    double cp = (hi - rp) - lo;                              // (+0 when k = 0; formed here, rp's last use, so that r takes rp's registers)
    SPA_KEEP(cp);
    const uint32_t sm = big ? 0u : 0x80000000u;
    const double r = SPA_MAKE(SPA_BITS_HI(rp) ^ sm, SPA_LO(rp));
    const double x2 = r * r;                                 // 2 hxs
    constexpr double q1 = 0.5 * Q1, q2 = 0.25 * Q2, q3 = 0.125 * Q3, q4 = 0.0625 * Q4, q5 = 0.03125 * Q5;
    double q1_v = q1, q2_v = q2, q3_v = q3, q4_v = q4, q5_v = q5;
    SPA_VREG(2, q1, 0xbf911111, 0x111110f4);
    const double R1 = 1.0 + x2 * q1_v, g2 = x2 * x2;         // g2 = 4 hxs^2
    SPA_VREG(4, q3, 0xbee4ce19, 0x9eaadbb7); SPA_VREG(4, q2, 0x3f3a01a0, 0x19fe5585);
    const double R2 = q2_v + x2 * q3_v, g4 = g2 * g2;        // R2 / 4, 16 hxs^4
    SPA_VREG(8, q5, 0xbe3afdb7, 0x6e09c32d); SPA_VREG(8, q4, 0x3e90cfca, 0x86e65239);
    const double R3 = q4_v + x2 * q5_v;                      // R3 / 16
    const double r1 = R1 + g2 * R2 + g4 * R3;
    constexpr double three = 3.0;
    double three_v = three;
    if constexpr ((SPA_VCONST & 512) == 0) SPA_SGPR(three_v);
    SPA_VREG(512, three, 0x40080000, 0x0);
    const double tt = __builtin_fma(-0.5, r1 * r, three_v);  // 3 - r1*hfx
    constexpr double six = 6.0;
    double six_v = six;
    SPA_VREG(16, six, 0x40180000, 0x0);
    const double den = six_v - r * tt;
    const double e2 = x2 * spa_div_r(r1 - tt, den, spa_recip(den));     // 2 e
    // s_expm1.c's k == 0 ending, x - (x*e - hxs), needs no branch of its own: with k = 0 the correction term c is +-0, so the general
    // e = (x*(e - c) - c) - hxs IS x*e - hxs bit for bit, and the ending is the difference r - e that the k = -1 and k <= -2 endings start from.
    const double c = SPA_MAKE(SPA_BITS_HI(cp) ^ sm, SPA_LO(cp));
    SPA_CENSUS(2);
    double e = r * __builtin_fma(0.5, e2, -c) - c;           // r*(e - c) - c
    e = __builtin_fma(-0.5, x2, e);                          // e -= hxs
    double t;
    if (big) {
        if (kk < 20) {
            SPA_CENSUS(4);
            const double tb = SPA_MAKE(0x3ff00000u - (0x200000u >> uint32_t(kk)), 0u);
            const double y = tb - (e - r);
            t = SPA_MAKE(SPA_BITS_HI(y) + (uint32_t(kk) << 20), SPA_LO(y));
        } else if (kk <= 56) {
            SPA_CENSUS(5);
            const double tc = SPA_MAKE(uint32_t(0x3ff - kk) << 20, 0u);
            const double y = (r - (e + tc)) + 1.0;
            t = SPA_MAKE(SPA_BITS_HI(y) + (uint32_t(kk) << 20), SPA_LO(y));
        } else {
            const double y = 1.0 - (e - r);
            t = SPA_MAKE(SPA_BITS_HI(y) + (uint32_t(kk) << 20), SPA_LO(y)) - 1.0;
        }
        SPA_KEEP(t);
    } else {
        t = r - e;                                           // k == 0: the result
        SPA_KEEP(t);
        if (kk == 1) {
            SPA_CENSUS(7);
            t = __builtin_fma(0.5, t, -0.5);                 // 0.5*(r - e) - 0.5: the product is exact
            SPA_KEEP(t);
        } else if (kk >= 2) {
            const double y = 1.0 + t;                        // 1 - (e - r)
            t = SPA_MAKE(SPA_BITS_HI(y) - (uint32_t(kk) << 20), SPA_LO(y)) - 1.0;
            SPA_CENSUS(8);
            SPA_KEEP(t);
        }
    }
    const double d2 = t + 2.0;
    const double rr = spa_recip(d2);
    double z;
    if (big) {
        z = spa_one_minus_2_over(d2, rr);
        SPA_CENSUS(9);
        SPA_KEEP(z);
    } else {
        z = spa_div_r(-t, d2, rr);
        SPA_CENSUS(10);
        SPA_KEEP(z);
    }
    double res = SPA_MAKE((SPA_BITS_HI(z) & 0x7fffffffu) | (jq & 0x80000000u), SPA_LO(z));     // z >= 0: one bit-field insert
    // s_tanh.c answers |x| < 2^-55 with x*(1 + x) = x. No branch for it here: down there k = 0, r = -A, r*r and everything multiplied by it
    // vanish against r, t = r, t + 2 = 2 and the quotient -t/2 IS 0.5*q, rounded once like the reference's own 0.5*Q when Q is denormal
    // (the host test sweeps every binade down to the smallest denormal).
    if (__builtin_expect(iq >= 0x40460000u, 0)) {
        res = SPA_MAKE(0x3ff00000u | (jq & 0x80000000u), 0u);
        SPA_CENSUS(11);
        if (q != q) res = q;                                 // s_tanh.c: one/x + one for a NaN is that NaN (the posterior array holds T before the first pass: a NaN must stay one)
    }
    return res;
}

// 2 * atanh(x) for |x| <= 1, with the decoder's clamp of +-1 to +-0.9999999 (ldpc_decoder_SPA.cc:150-156)
SPA_FN double spa_atanh_x2(double x) {
    constexpr double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    constexpr double Lp1 = 6.666666666666735130e-01, Lp2 = 3.999999999940941908e-01, Lp3 = 2.857142874366239149e-01,
                     Lp4 = 2.222219843214978396e-01, Lp5 = 1.818357216161805012e-01, Lp6 = 1.531383769920937332e-01,
                     Lp7 = 1.479819860511658591e-01;
    if (__builtin_expect(spa_fabs(x) == 1.0, 0)) {
        x = SPA_MAKE((SPA_BITS_HI(x) & 0x80000000u) | 0x3fefffffu, 0xca501acbu);      // +-0.9999999
        SPA_CENSUS(13);
    }
    SPA_KEEP(x);                       // the clamp rewrites x itself; |x| below stays a source modifier, never a register pair of its own
    const uint32_t jx = SPA_BITS_HI(x);
    const double xa = spa_fabs(x);
    SPA_CENSUS(12);
    const double d1 = 1.0 - xa;
    const double r1 = spa_recip(d1);
    double yh;                         // y / 2, y = 2x/(1-x) resp. 2x + 2x*x/(1-x)
    if (xa < 0.5) {
        yh = xa + spa_div_r(xa * xa, d1, r1);
        SPA_CENSUS(14);
        SPA_KEEP(yh);
    } else {
        yh = spa_div_r(xa, d1, r1);
        SPA_CENSUS(15);
        SPA_KEEP(yh);
    }
    // log1p(y)
    double fh, l;                      // fh = f / 2
    const bool direct = int32_t(SPA_BITS_HI(yh)) < 0x3FDA827A - 0x00100000;      // y's high word < 0x3FDA827A
    // The normalised lanes finish with y first (the correction term c = (1 + y - u)/u, exactly as s_log1p.c forms it) and
    // then turn it into f in place; the direct lanes' f is y itself - no lane copies a register pair. Everything only the
    // normalised lanes need (u = 1 + y and what is cut out of its high word) is computed under their branch: a wavefront whose
    // checks are all weak (every y below 0.41421 - the rule far from convergence) skips it.
    double c;
    uint32_t hadd, hm;
    SPA_UNDEF(c); SPA_UNDEF(hadd); SPA_UNDEF(hm);
    fh = yh;
    if (!direct) {
        const double u = __builtin_fma(2.0, yh, 1.0);        // 1 + y
        SPA_CENSUS(16);
        // s_log1p.c normalises u = 1 + y to [sqrt(2)/2, sqrt(2)): k = exponent, + 1 when the top 20 mantissa bits are >= 0x6a09e.
        // Adding 0x100000 - 0x6a09e to the high word carries into the exponent field in exactly that case, so one addition
        // gives k ((hadd >> 20) - 1023), the normalised high word ((hadd & 0xfffff) + 0x3fe6a09e = hu | 0x3ff00000 resp.
        // hu | 0x3fe00000) and the |f| < 2^-20 test (hu == 0 resp. (0x100000 - hu) >> 2 == 0 <=> (hadd & 0xfffff) in 0x95f5f..0x95f62)
        hadd = SPA_BITS_HI(u) + 0x95f62u;
        hm = hadd & 0x000fffffu;
        // The correction term c = (rounding error of u = fl(1 + y)) / u. s_log1p.c forms the error as 1 - (u - y) when u >= 2 and as y - (u - 1)
        // otherwise; both are Dekker's error-free difference, and y - (u - 1) is exact for u >= 2 as well (u - 1 is exact for 2 <= u < 2^53, and
        // the difference to y is the representable rounding error itself), so ONE form serves every normalised lane - the same value, bit for
        // bit, without the two execution-mask regions (round 6: -1.6 % at the waterfall point, -1.9 % on mode 0; the host test sweeps the
        // u ~ 2^k switch points against libm).
        c = __builtin_fma(2.0, yh, -(u - 1.0));              // y - (u - 1)
        SPA_CENSUS(18);
        SPA_KEEP(c);
        c = spa_div_r(c, u, spa_recip(u));
        fh = SPA_MAKE(hm + 0x3fd6a09eu, SPA_LO(u)) - 0.5;    // (normalised u)/2 - 1/2
        SPA_KEEP(fh);
        SPA_KEEP(c);
    }
    const double hfsqh = fh * fh;                            // hfsq / 2  (hfsq = 0.5 f f)
    const double d3 = __builtin_fma(2.0, fh, 2.0);           // 2 + f
    const double sh = spa_div_r(fh, d3, spa_recip(d3));      // s / 2
    const double zq = sh * sh;                               // z / 4
    constexpr double l1 = 8 * Lp1, l2 = 32 * Lp2, l3 = 128 * Lp3, l4 = 512 * Lp4, l5 = 2048 * Lp5, l6 = 8192 * Lp6, l7 = 32768 * Lp7;
    double l1_v = l1, l2_v = l2, l3_v = l3, l4_v = l4, l5_v = l5, l6_v = l6, l7_v = l7;
    SPA_VREG(32, l1, 0x40155555, 0x55555593);
    const double R1 = zq * l1_v, z2 = zq * zq;                           // 2 z Lp1 ; z^2 / 16
    SPA_VREG(64, l3, 0x40424924, 0x94229359); SPA_VREG(64, l2, 0x40299999, 0x9997fa04);
    const double R2 = l2_v + zq * l3_v, z4 = z2 * z2;                    // 32 (Lp2 + z Lp3) ; z^4 / 256
    SPA_VREG(128, l5, 0x40774664, 0x96cb03de); SPA_VREG(128, l4, 0x405c71c5, 0x1d8e78af);
    const double R3 = l4_v + zq * l5_v, z6 = z4 * z2;                    // 512 (Lp4 + z Lp5) ; z^6 / 4096
    SPA_VREG(256, l7, 0x40b2f112, 0xdf3e5244); SPA_VREG(256, l6, 0x40939a09, 0xd078c69f);
    const double R4 = l6_v + zq * l7_v;                                  // 8192 (Lp6 + z Lp7)
    const double R = R1 + z2 * R2 + z4 * R3 + z6 * R4;                   // 2 R
    const double sr = sh * __builtin_fma(4.0, hfsqh, R);     // s * (hfsq + R)
    if (direct) {
        l = __builtin_fma(2.0, fh, -__builtin_fma(2.0, hfsqh, -sr));     // f - (hfsq - s*(hfsq + R))
        SPA_CENSUS(19);
        SPA_KEEP(l);
    } else {
        const double dk = double(int32_t(hadd >> 20) - 1023);
        SPA_CENSUS(20);
        if (__builtin_expect(hm - 0x95f5fu < 4u, 0)) {
            const double f = fh + fh, hfsq = hfsqh + hfsqh;
            SPA_CENSUS(21);
            if (f == 0.0) {
                l = dk * ln2_hi + (c + dk * ln2_lo);
            } else {
                const double Rz = hfsq * (1.0 - 0.66666666666666666 * f);
                l = dk * ln2_hi - ((Rz - (dk * ln2_lo + c)) - f);
            }
        } else {
            const double w = __builtin_fma(2.0, hfsqh, -(sr + (dk * ln2_lo + c)));   // hfsq - (s*(hfsq + R) + (k*ln2_lo + c))
            l = __builtin_fma(dk, ln2_hi, -__builtin_fma(-2.0, fh, w));              // k*ln2_hi - (w - f): k*ln2_hi is exact
        }
        SPA_KEEP(l);
    }
    double res = SPA_MAKE((SPA_BITS_HI(l) & 0x7fffffffu) | (jx & 0x80000000u), SPA_LO(l));     // l > 0: 2 * (+-0.5 * l) is one bit-field insert
    if (__builtin_expect(xa < 0x1.0p-28, 0)) { res = x + x; SPA_CENSUS(22); }
    return res;
}

// The decoder's calls (ldpc.hip). Three of the two routines' answers need none of their arithmetic, and a wavefront whose lanes ALL get one
// of them skips the evaluation - like the reference's libm, which returns at once:
//  * tanh, |x| >= 22 -> +-1 (s_tanh.c). The zero-forcing modes hand the decoder +-Inf (their pilots equalise onto themselves, the measured
//    variance is exactly 0 and every LLR is (d1 - d0)/0): every wavefront of theirs, in every iteration of a frame that does not pass its parity
//    checks at once. (NaN lanes make the wavefront take the evaluation.)
//  * atanh, the decoder's clamp of +-1 (ldpc_decoder_SPA.cc:150-156): 2 atanh(+-0.9999999) is ONE constant, 0x1.0cfad9b61ff69p+4 (what
//    spa_atanh_x2 returns for it: tests/test_spa_math.py). Hard inputs make every product of every wavefront +-1.
//  * atanh, |x| < 2^-28 -> x (e_atanh.c). A check of high degree far from convergence multiplies several dozen small tanh values: every
//    product of the wavefront is down there (rate 14/16, noise only: all of them; profiles/r05_spa_branch_census.txt).
// (lanes: the wavefront's lanes that count - ldpc.hip's padding lanes compute along on dummy values and must not veto a shortcut)
SPA_FN double spa_tanh_half_wave(double q, unsigned long long lanes = ~0ull) {
    if (SPA_ALL_OF(lanes, spa_fabs(q) >= 44.0)) { SPA_CENSUS(24); return SPA_MAKE(0x3ff00000u | (SPA_BITS_HI(q) & 0x80000000u), 0u); }
    return spa_tanh_half(q);
}
SPA_FN double spa_atanh_x2_wave(double x, unsigned long long lanes = ~0ull) {
    const bool unit = spa_fabs(x) == 1.0, tiny = spa_fabs(x) < 0x1.0p-28;
    if (SPA_ALL_OF(lanes, unit || tiny)) {
        SPA_CENSUS(25);
        return unit ? SPA_MAKE((SPA_BITS_HI(x) & 0x80000000u) | 0x4030cfadu, 0x9b61ff69u) : x + x;
    }
    return spa_atanh_x2(x);
}

// the plain forms (tests, probes): tanh(x) = tanh(0.5 * 2x), atanh(x) = 0.5 * (2 atanh(x)); both scalings are exact
SPA_FN double spa_tanh(double x) { return spa_tanh_half(x + x); }
SPA_FN double spa_atanh(double x) { return 0.5 * spa_atanh_x2(x); }
