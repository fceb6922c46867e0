// Scalar building blocks of the front-end arithmetic shared by the fused kernel (frontend.hip) and the per-method
// stage kernels (stages.hip): the reference's complex division, linear interpolation and angle, operation for operation.
#pragma once
#include "fft256.h"   // c2
#include "glibc_trig.h"   // atan / sincos as the reference platform's libm evaluates them, bit for bit

// libgcc (GCC 11) __divdc3 main path
__device__ __forceinline__ c2 cdiv(c2 n, c2 d) {
    const double a = n.re, b = n.im, c = d.re, dd = d.im;
    double x, y;
    if (fabs(c) < fabs(dd)) {
        const double ratio = c / dd, denom = (c * ratio) + dd;
        x = ((a * ratio) + b) / denom;
        y = ((b * ratio) - a) / denom;
    } else {
        const double ratio = dd / c, denom = (dd * ratio) + c;
        x = ((b * ratio) + a) / denom;
        y = (b - (a * ratio)) / denom;
    }
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

__device__ __forceinline__ double get_angle(c2 v) {
    double theta = 0;
    if (v.re == 0) theta = M_PI / 2;
    else if (v.re > 0) theta = gl_atan(v.im / v.re);
    else if (v.re < 0 && v.im >= 0) theta = gl_atan(v.im / v.re) + M_PI;
    else if (v.re < 0 && v.im < 0) theta = gl_atan(v.im / v.re) - M_PI;
    return theta;
}

