// Scalar building blocks of the front-end arithmetic shared by the fused kernel (frontend.hip) and the per-method
// stage kernels (stages.hip): the reference's complex division, linear interpolation and angle, operation for operation.
#pragma once
#include "fft256.h"   // c2
#include "glibc_trig.h"   // atan / sincos as the reference platform's libm evaluates them, bit for bit

// libgcc (GCC 11) __divdc3 main path. The two cases (|c| < |d| or not) are the same five operations on swapped operands, so the
// operands are selected and the operations done once: lanes of a wavefront take either case at random, and as two branches every
// wavefront paid for both (six divisions instead of three).
__device__ __forceinline__ c2 cdiv(c2 n, c2 d) {
    const double a = n.re, b = n.im, c = d.re, dd = d.im;
    const bool sw = fabs(c) < fabs(dd);
    const double p = sw ? c : dd, q = sw ? dd : c;
    const double ratio = p / q, denom = (p * ratio) + q;        // c / d, (c * ratio) + d   or   d / c, (d * ratio) + c
    const double u = sw ? a : b, v = sw ? b : a;
    const double x = ((u * ratio) + v) / denom;                // ((a * ratio) + b) / denom   or   ((b * ratio) + a) / denom
    const double t = v * ratio;
    const double y = (sw ? t - a : b - t) / denom;             // ((b * ratio) - a) / denom   or   (b - (a * ratio)) / denom
    return {x, y};
}

// interpolate_linear (complex): a + (b-a)*(x-a_x)/(b_x-a_x), component-wise scalings
__device__ __forceinline__ c2 lerp(c2 a, double ax, c2 b, double bx, double x) {
    const double m = x - ax, q = bx - ax;
    c2 t = {(b.re - a.re) * m, (b.im - a.im) * m};
    t.re = t.re / q;
    t.im = t.im / q;
    return {a.re + t.re, a.im + t.im};
}

// get_angle (misc.cc:34-56): every branch that calls atan calls it on v.im / v.re, so it is evaluated once
__device__ __forceinline__ double get_angle(c2 v) {
    const double a = gl_atan(v.im / v.re);
    double theta = 0;
    if (v.re == 0) theta = M_PI / 2;
    else if (v.re > 0) theta = a;
    else if (v.re < 0 && v.im >= 0) theta = a + M_PI;
    else if (v.re < 0 && v.im < 0) theta = a - M_PI;
    return theta;
}
