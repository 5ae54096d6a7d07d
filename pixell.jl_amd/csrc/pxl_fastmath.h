// pxl_fastmath.h -- atan2, asin, sincos and rsqrt for the Gnomonic evaluators; included by pxl_kernels.hip and, on the host, by
// tests/native/fastmath_check.cpp (which measures them against long double libm: tests/test_fastmath.py).
//
// Why not the device libm: the Gnomonic evaluators are bound by FP64 instruction issue, and ocml's atan2 / asin / sincos cost
// well over a hundred wave instructions each (rational kernels with a second division, IEEE divisions and square roots with their
// scaling steps, selects of 64-bit constants), its sincos ~65 (double-double reduction, quadrant selects).
// The paths that use these functions are tolerance-checked against the oracle (glibc) anyway -- the reference's own bar for its
// Gnomonic code is an L1 bound against wcslib (test_geometry.jl:116-119) -- so what is needed is a couple of ulp, cheaply:
//   pxl_fm_atan2   one division (reciprocal seed + two Newton steps), an 11-term polynomial on |t| <= tan(pi/8), two further reduction centres (1/2 and 1) chosen so
//                  that the reduced angle is at most 0.28 of the result (its error is not amplified); <= 2 ulp measured
//   pxl_fm_asin    one 13-term polynomial for both halves (|v| <= 1/2 directly, else pi/2 - 2 asin(sqrt((1 - |v|)/2)) with the
//                  square root's residual carried along); <= 2 ulp measured
//   pxl_fm_sincos  Cody-Waite reduction by pi/2 in two 53-bit parts (first step exact, second rounded once), 7- and 6-term
//                  kernels, quadrant step = one swap and two sign flips; <= 1.5 ulp of max(|sin|, |cos|) (absolute) for |x| <= 2^19 pi/2, refused beyond (the
//                  caller then uses the library).  A first version that carried the reduction's tail into both kernels and chose
//                  with 64-bit selects was 9 % SLOWER than the library's; this one is 16 % faster (sky2pix 57 -> 66 %)
//   pxl_fm_rsqrt   seed + one third-order step; <= 1 ulp
// Every product-sum is an explicit fma (the translation unit is compiled with -ffp-contract=off).  Coefficients:
// tools/gen_fastmath_coeffs.py (Chebyshev fits at 60 digits).
#pragma once
#include "pxl_fastmath_coeffs.h"

#ifndef PXL_FM_HD
#define PXL_FM_HD __host__ __device__ inline
#endif

// a * b + c with c a compile-time constant.  On the device the constant is read from a scalar register pair by a VOP3 fma:
// left to itself the compiler copies every coefficient into a vector register first (v_mov_b64 + v_fmac per Horner step).
PXL_FM_HD double pxl_fm_fma_k(double a, double b, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return r;
#else
    return __builtin_fma(a, b, c);
#endif
}
// a * k + c with k a compile-time constant
PXL_FM_HD double pxl_fm_kfma(double a, double k, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(k), "v"(c));
    return r;
#else
    return __builtin_fma(a, k, c);
#endif
}

template <int N>
PXL_FM_HD double pxl_fm_horner(const double (&c)[N], double z) {
    double p = c[N - 1];
#pragma unroll
    for (int i = N - 2; i >= 0; --i) p = pxl_fm_fma_k(p, z, c[i]);
    return p;
}

// the hardware's reciprocal and reciprocal-square-root seeds (v_rcp_f64, v_rsq_f64).  On the host: the exact value cut to 24
// bits, so that the harness exercises the refinements below with seeds no better than the device's
#if !defined(__HIP_DEVICE_COMPILE__)
static inline double pxl_fm_cut24(double v) {
    unsigned long long b;
    __builtin_memcpy(&b, &v, 8);
    b &= ~((1ULL << 29) - 1);
    __builtin_memcpy(&v, &b, 8);
    return v;
}
#endif
PXL_FM_HD double pxl_fm_rsq_seed(double w) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rsq(w);
#else
    return pxl_fm_cut24(1.0 / __builtin_sqrt(w));
#endif
}
PXL_FM_HD double pxl_fm_rcp_seed(double w) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcp(w);
#else
    return pxl_fm_cut24(1.0 / w);
#endif
}

// 1/sqrt(u) for u in [2^-1000, 2^1000] to ~1 ulp: seed y0 (relative error e0 <= 2^-20), then y0 (1 + e/2 + 3 e^2/8), e = 1 - u y0^2
// Domain: finite u > 0.  +Inf gives NaN (Inf * 0 in the refinement) -- deliberately: the evaluators multiply the result by a
// numerator that overflows together with u, and a quiet 0 there would turn into a wrong finite angle.
PXL_FM_HD double pxl_fm_rsqrt(double u) {
    const double y0 = pxl_fm_rsq_seed(u);
    const double e = __builtin_fma(-(u * y0), y0, 1.0);
    return __builtin_fma(y0 * e, __builtin_fma(0.375, e, 0.5), y0);
}

// atan2(y, x), IEEE special cases included (signed zeros, infinities, NaN).  Branch-free, and the choices are made with 0 / 1
// flags in arithmetic (flag * constant is exact) rather than with selects of 64-bit constants, which would each cost two
// v_cndmask and a vector register pair per constant.
// TAME (compile time): the caller guarantees x > 0 finite and max(|x|, |y|) in [2^-764, 2^764] with y finite (it has tested a
// whole wave, pxl_fm_atan2_is_tame): the infinity, scaling, zero-denominator, negative-x and NaN steps are left out -- a sixth of
// the instructions.  Same bits as the general form on such arguments.
PXL_FM_HD bool pxl_fm_atan2_is_tame(double y, double x) {
    const double m = __builtin_fmax(__builtin_fabs(x), __builtin_fabs(y));
    return x > 0.0 && m >= 0x1p-764 && m <= 0x1p+764 && y == y;          // (NaN x fails x > 0; a NaN y would be dropped by fmax)
}
template <bool TAME = false>
PXL_FM_HD double pxl_fm_atan2(double y, double x) {
    constexpr double Q[PXL_FM_ATAN_Q_N] = PXL_FM_ATAN_Q;
    const double ax = __builtin_fabs(x), ay = __builtin_fabs(y);
    const double fs = (ay > ax) ? 1.0 : 0.0;                     // swap: the angle is measured from the y axis
    double mx = __builtin_fmax(ax, ay), mn = __builtin_fmin(ax, ay);
    if (!TAME) {
        const bool mxinf = mx == __builtin_inf();
        mn = mxinf ? ((mn == __builtin_inf()) ? 1.0 : 0.0) : mn;
        mx = mxinf ? 1.0 : mx;
        // bring the pair into [2^-764, 2^764] (exact): the sums below cannot overflow, nothing is subnormal, and the quotient can
        // be formed from the reciprocal seed without the scaling steps of an IEEE division
        const double sc = mx > 0x1p+764 ? 0x1p-260 : (mx < 0x1p-764 ? 0x1p+260 : 1.0);
        mn *= sc;
        mx *= sc;
    }
    // atan(mn / mx) = o + atan(t) with the centre c (o = atan c) nearest below the ratio among 0, 1/2, 1:
    //   t = (mn - c mx) / (mx + c mn);  the numerators mn - mx/2 and mn - mx are exact (Sterbenz) in their intervals
    const bool i1 = mn > PXL_FM_TAN_PIO8 * mx;          // ratio above tan(pi/8)
    const bool i2 = mn > 0.75 * mx;                     // ratio in (3/4, 1]: centre 1
    const double f2 = i2 ? 1.0 : 0.0;
    const double f1 = (i1 && !i2) ? 1.0 : 0.0;          // ratio in (tan(pi/8), 3/4]: centre 1/2
    const double c = pxl_fm_kfma(f1, 0.5, f2);
    const double o_hi = pxl_fm_kfma(f1, PXL_FM_ATAN_HALF_HI, f2 * PXL_FM_PIO4_HI);
    const double o_lo = pxl_fm_kfma(f1, PXL_FM_ATAN_HALF_LO, f2 * PXL_FM_PIO4_LO);
    const double num = __builtin_fma(-c, mx, mn);
    double den = __builtin_fma(c, mn, mx);
    if (!TAME) den = (den == 0.0) ? 1.0 : den;          // atan2(+-0, +-0): t = 0
    double rd = pxl_fm_rcp_seed(den);
    rd = __builtin_fma(__builtin_fma(-den, rd, 1.0), rd, rd);
    rd = __builtin_fma(__builtin_fma(-den, rd, 1.0), rd, rd);
    const double t0 = num * rd;
    const double t = __builtin_fma(__builtin_fma(-den, t0, num), rd, t0);      // num / den to <= 1 ulp
    const double z = t * t;
    const double r = __builtin_fma(t * z, pxl_fm_horner(Q, z), t);         // atan(t)
    double a = o_hi + (r + o_lo);                       // the octant's angle, in [0, pi/4]
    // octant -> quadrant: swap mirrors about pi/4 (pi/2 - a), a negative x about pi/2 (pi - a): K + (+-a + K_lo), K = flag * constant
    a = pxl_fm_kfma(fs, PXL_FM_PIO2_1, pxl_fm_kfma(fs, PXL_FM_PIO2_2, __builtin_fma(-2.0, fs, 1.0) * a));
    if (TAME) return __builtin_copysign(a, y);
    const double fx = __builtin_signbit(x) ? 1.0 : 0.0;
    a = pxl_fm_kfma(fx, PXL_FM_PI_HI, pxl_fm_kfma(fx, PXL_FM_PI_LO, __builtin_fma(-2.0, fx, 1.0) * a));
    a = __builtin_copysign(a, y);
    return (x != x || y != y) ? x + y : a;
}

// w with a tiny negative value, w in [-2^-50, -0], replaced by (effectively) +0: one integer compare and one select on the high
// word.  Positive values, larger negative ones and NaN pass unchanged.
PXL_FM_HD double pxl_fm_lift_tiny_negative(double w) {
    unsigned long long b;
    __builtin_memcpy(&b, &w, 8);
    const int hi = (int)(unsigned)(b >> 32);
    // as a signed word: every non-negative double has hi >= 0; the negative ones run from INT_MIN (-0) upwards with their magnitude,
    // and 0xBCD... is the high word of -2^-50
    const unsigned hi2 = hi <= (int)0xBCD00000u ? 0u : (unsigned)hi;
    b = ((unsigned long long)hi2 << 32) | (b & 0xFFFFFFFFull);          // the low word may stay: below 2^-1022 either way
    __builtin_memcpy(&w, &b, 8);
    return w;
}

// asin(v); NaN for |v| > 1 + 2^-49, and SATURATING at +-pi/2 for 1 < |v| <= 1 + 2^-49: the evaluators' argument is a quotient that is
// at most 1 mathematically but is formed from three rounded factors (numerator, reciprocal square root, their product), so it
// overshoots 1 by up to 4.4e-16 for pixels on a celestial pole (2.26 M of 20 M patch centres) -- those are the pole,
// not an error.  HALF (compile time): 0 = any v (both halves evaluated, one selected); 1 = the caller guarantees
// |v| <= 1/2, 2 = |v| > 1/2 or NaN (it has tested a whole wave): only that half is evaluated -- the same operations, the same bits.
template <int HALF = 0>
PXL_FM_HD double pxl_fm_asin(double v) {
    constexpr double R[PXL_FM_ASIN_R_N] = PXL_FM_ASIN_R;
    const double av = __builtin_fabs(v);
    if (HALF == 1) {
        const double z = v * v;
        return __builtin_copysign(av + (av * z) * pxl_fm_horner(R, z), v);
    }
    const bool big = HALF == 2 || av > 0.5;
    // big: asin(av) = pi/2 - 2 asin(sqrt(w)), w = (1 - av) / 2 (exact for av in (1/2, 1], and then in [2^-54, 1/4) or 0)
    const double w = pxl_fm_lift_tiny_negative(__builtin_fma(-0.5, av, 0.5));
    // sqrt(w) = b + blo: one Goldschmidt step from the seed, then two corrections by the exact residual w - s^2
    const double y0 = pxl_fm_rsq_seed(w + 0x1p-200);                       // w itself for w >= 2^-54; finite for w = 0 (|v| = 1)
    const double h = 0.5 * y0;
    const double s0 = w * y0;
    const double s1 = __builtin_fma(s0, __builtin_fma(-s0, h, 0.5), s0);
    const double b1 = __builtin_fma(__builtin_fma(-s1, s1, w), h, s1);
    const double blo = __builtin_fma(-b1, b1, w) * h;
    const double z = big ? w : v * v;
    const double b = big ? b1 : av;
    const double p = (b * z) * pxl_fm_horner(R, z);                        // asin(b) - b
    // pi/2 - 2 (b + blo + p): the leading difference with its rounding error (fast two-sum: pi/2 >= 2 b)
    const double b2 = b1 + b1;
    const double u = PXL_FM_PIO2_1 - b2;
    const double uerr = (PXL_FM_PIO2_1 - u) - b2;
    const double rbig = u + (uerr - (((blo + p) + (blo + p)) - PXL_FM_PIO2_2));
    if (HALF == 2) return __builtin_copysign(rbig, v);
    const double rsmall = av + p;
    return __builtin_copysign(big ? rbig : rsmall, v);
}

// v with the sign flipped when `neg` is nonzero in bit 1 (neg = q & 2, or (q + 1) & 2): one shift and one xor on the high word
PXL_FM_HD double pxl_fm_flip_if_bit1(double v, int q) {
    unsigned long long b;
    __builtin_memcpy(&b, &v, 8);
    b ^= (unsigned long long)((unsigned)q & 2u) << 62;
    __builtin_memcpy(&v, &b, 8);
    return v;
}

// sin and cos of x for |x| <= 2^19 pi/2 (returns true); false, nothing written, beyond that and for NaN / Inf: the caller uses
// the library.  Cody-Waite reduction by pi/2 in two 53-bit parts -- the first step is exact (|n| < 2^20: the difference is a
// multiple of 2^-53 below 1 in magnitude), the second is rounded once, which is the whole reduction error: 0.5 ulp of the reduced
// argument (the third part of pi/2, n * 1e-33, is dropped) -- then the two kernels on |r| <= pi/4 and a quadrant step made of one
// swap and two sign flips.  Error bound: <= 1.5 ulp OF max(|sin|, |cos|), i.e. an ABSOLUTE 1.7e-16 -- not a relative bound on the
// small component next to a multiple of pi/2: there the dropped third part of pi/2 (n * 1.5e-33) and the rounding of the second step
// (2^-53 |n| pi/2 relative to a reduced argument that can be as small as 3e-16 at n ~ 5e5) leave an absolute error of ~1e-27 in a
// value of ~3e-16, 1e-12 relative.  The evaluators multiply these sines and cosines into direction cosines of order 1,
// so the absolute bound is the one that matters; tests/native/fastmath_check.cpp checks it at the worst-case arguments near
// k pi/2 as an absolute bound.  About 40 instructions against the library's ~65 (half of those are its reduction's double-double
// arithmetic and the selects of the quadrant logic).
PXL_FM_HD bool pxl_fm_sincos(double x, double* sn, double* cs) {
    constexpr double S[PXL_FM_SIN_S_N] = PXL_FM_SIN_S;
    constexpr double Cc[PXL_FM_COS_C_N] = PXL_FM_COS_C;
    if (!(__builtin_fabs(x) <= 823549.0)) return false;
    const double n = __builtin_rint(x * PXL_FM_2OPI);
    const double r = pxl_fm_kfma(-n, PXL_FM_PIO2_2, pxl_fm_kfma(-n, PXL_FM_PIO2_1, x));
    const double z = r * r;
    const double sv = __builtin_fma(r * z, pxl_fm_horner(S, z), r);         // r + r z S(z)
    // 1 - z/2 + z^2 C(z), the leading difference with its rounding error
    const double hz = 0.5 * z;
    const double one_m = 1.0 - hz;
    const double cv = one_m + (((1.0 - one_m) - hz) + (z * z) * pxl_fm_horner(Cc, z));
    // quadrant q = n mod 4: (sin, cos) = (s, c), (c, -s), (-s, -c), (-c, s)
    const int q = (int)n;
    const bool odd = q & 1;
    *sn = pxl_fm_flip_if_bit1(odd ? cv : sv, q);
    *cs = pxl_fm_flip_if_bit1(odd ? sv : cv, q + 1);
    return true;
}
