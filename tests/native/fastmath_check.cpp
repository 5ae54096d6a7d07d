// fastmath_check.cpp -- accuracy of pixell.jl_amd/csrc/pxl_fastmath.h on the host against long double libm (x87: 64-bit
// significand, i.e. 2^-11 of a double's ulp).  Built and run by tests/test_fastmath.py; prints one JSON object.
//   g++ -O2 -std=c++17 -ffp-contract=off -I pixell.jl_amd/csrc tests/native/fastmath_check.cpp -o /tmp/fastmath_check
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#define PXL_FM_HD static inline
#include "pxl_fastmath.h"

static double ulp_of(long double v) {            // spacing of doubles at |v|
    double d = std::fabs((double)v);
    if (d < 2.2250738585072014e-308) return 4.9406564584124654e-324;
    int e;
    std::frexp(d, &e);
    return std::ldexp(1.0, e - 53);
}
static double err_ulp(double got, long double want) {
    if (std::isnan(got) || std::isnan((double)want)) return (std::isnan(got) && std::isnan((double)want)) ? 0.0 : 1e30;
    return (double)(fabsl((long double)got - want) / (long double)ulp_of(want));
}
static bool same_bits(double a, double b) { return std::memcmp(&a, &b, 8) == 0 || (std::isnan(a) && std::isnan(b)); }

int main(int argc, char** argv) {
    const long n = argc > 1 ? atol(argv[1]) : 4000000;
    std::mt19937_64 rng(20261004);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    double e_atan2 = 0, e_asin = 0;
    double w_atan2[2] = {0, 0}, w_asin = 0, e_rsqrt = 0, w_rsqrt = 0;
    long n_tame = 0, tame_bad = 0, half_bad = 0;
    for (long k = 0; k < n; ++k) {
        // atan2: angles uniform on the circle (and clustered at the octant / interval boundaries), radii over 600 binades
        double ang = (k & 1) ? (U(rng) * 2 - 1) * M_PI : std::round(U(rng) * 64) * (M_PI / 32) + (U(rng) - 0.5) * 1e-6;
        if ((k & 7) == 3) ang = std::atan(0.75) + (U(rng) - 0.5) * 1e-9;
        if ((k & 7) == 5) ang = M_PI / 8 + (U(rng) - 0.5) * 1e-9;
        double rad = std::exp2((U(rng) - 0.5) * 600);
        double y = rad * std::sin(ang), x = rad * std::cos(ang);
        const double got = pxl_fm_atan2(y, x);
        double e = err_ulp(got, atan2l((long double)y, (long double)x));
        if (e > e_atan2) { e_atan2 = e; w_atan2[0] = y; w_atan2[1] = x; }
        // the TAME form gives the same bits wherever its precondition holds, and the precondition test refuses everything else
        if (pxl_fm_atan2_is_tame(y, x)) { ++n_tame; if (!same_bits(pxl_fm_atan2<true>(y, x), got)) ++tame_bad; }
        else if (x > 0 && std::isfinite(y) && std::fmax(std::fabs(x), std::fabs(y)) >= 0x1p-700 && std::fmax(std::fabs(x), std::fabs(y)) <= 0x1p+700) ++tame_bad;
        // asin: uniform, clustered at 0, 1/2 and 1
        double v = U(rng) * 2 - 1;
        if ((k & 3) == 1) v = std::copysign(0.5 + (U(rng) - 0.5) * 1e-3, v);
        if ((k & 3) == 2) v = std::copysign(1.0 - U(rng) * U(rng) * 1e-2, v);
        if ((k & 15) == 7) v = std::exp2(-U(rng) * 60) * (v < 0 ? -1 : 1);
        const double gasin = pxl_fm_asin(v);
        e = err_ulp(gasin, asinl((long double)v));
        if (e > e_asin) { e_asin = e; w_asin = v; }
        // the one-half forms give the bits of the general one
        if (!same_bits(std::fabs(v) <= 0.5 ? pxl_fm_asin<1>(v) : pxl_fm_asin<2>(v), gasin)) ++half_bad;
        // rsqrt: the evaluators call it on 1 + X^2 + Y^2; here over 200 binades
        {
            double uu = std::exp2((U(rng) - 0.5) * 200);
            if (k & 1) uu = 1.0 + U(rng) * 3;
            double er = err_ulp(pxl_fm_rsqrt(uu), 1.0L / sqrtl((long double)uu));
            if (er > e_rsqrt) { e_rsqrt = er; w_rsqrt = uu; }
        }
    }
    // sincos: |x| <= 8 (the evaluators' range), near multiples of pi/2, and up to the limit of the fast path
    double e_sin = 0, e_cos = 0, e_sin_big = 0, e_cos_big = 0, w_sin = 0, w_cos = 0;
    for (long k = 0; k < n; ++k) {
        double xs = (U(rng) * 2 - 1) * 8;
        if ((k & 3) == 1) xs = std::round((U(rng) * 2 - 1) * 16) * (M_PI / 2) + (U(rng) - 0.5) * std::exp2(-U(rng) * 30);
        double sn, cs;
        if (!pxl_fm_sincos(xs, &sn, &cs)) { printf("{\"error\": \"fast path refused %g\"}\n", xs); return 1; }
        double es = err_ulp(sn, sinl((long double)xs)), ec = err_ulp(cs, cosl((long double)xs));
        if (es > e_sin) { e_sin = es; w_sin = xs; }
        if (ec > e_cos) { e_cos = ec; w_cos = xs; }
        double xl = (U(rng) * 2 - 1) * 823549.0;
        if (!pxl_fm_sincos(xl, &sn, &cs)) { printf("{\"error\": \"fast path refused %g\"}\n", xl); return 1; }
        e_sin_big = std::max(e_sin_big, err_ulp(sn, sinl((long double)xl)));
        e_cos_big = std::max(e_cos_big, err_ulp(cs, cosl((long double)xl)));
    }
    // the worst arguments for the two-part reduction: the doubles nearest to k pi/2 (and their neighbours), every k up to the limit
    // of the fast path.  The small component there is ~1e-16..1e-11 and only an ABSOLUTE bound holds for it (pxl_fastmath.h).
    double e_near_abs = 0, w_near = 0;
    {
        const long double pio2l = 1.57079632679489661923132169163975144L;
        for (long k = 1; k <= 524287; ++k) {
            const double xc = (double)(k * pio2l);
            for (double xs : {xc, std::nextafter(xc, 0.0), std::nextafter(xc, 1e9), -xc}) {
                double sn, cs;
                if (!pxl_fm_sincos(xs, &sn, &cs)) { printf("{\"error\": \"fast path refused %g\"}\n", xs); return 1; }
                const double ea = (double)std::max(fabsl((long double)sn - sinl((long double)xs)), fabsl((long double)cs - cosl((long double)xs)));
                if (ea > e_near_abs) { e_near_abs = ea; w_near = xs; }
            }
        }
    }
    // special cases: bit-for-bit what libm returns
    const double inf = INFINITY, nan = NAN;
    const double sp[] = {0.0, -0.0, 1.0, -1.0, inf, -inf, nan, 5e-324, -5e-324, 1e308, -1e308, 0.5, 0.75, 2.0};
    int special_bad = 0;
    for (double y : sp)
        for (double x : sp) {
            double g = pxl_fm_atan2(y, x), w = std::atan2(y, x);
            if (!(same_bits(g, w) || err_ulp(g, atan2l((long double)y, (long double)x)) <= 1.0) || std::signbit(g) != std::signbit(w)) {
                if (!(std::isnan(g) && std::isnan(w))) { ++special_bad; fprintf(stderr, "atan2(%g, %g) = %a, libm %a\n", y, x, g, w); }
            }
        }
    // saturation band: 1 < |v| <= 1 + 2^-49 is the pole (a rounded quotient of at most 1), beyond it NaN as in libm
    for (double v : {1.0000000000000002, -1.0000000000000002, 1.0000000000000004, 1.0 + 0x1p-49, -(1.0 + 0x1p-49)}) {
        const double want = std::copysign(std::asin(1.0), v);
        for (double g : {pxl_fm_asin(v), pxl_fm_asin<2>(v)})
            if (!same_bits(g, want)) { ++special_bad; fprintf(stderr, "asin(%a) = %a, expected the pole %a\n", v, g, want); }
    }
    for (double v : {1.0 + 0x1p-48, -(1.0 + 0x1p-48), 1.00000001, -1.00000001, 2.0})
        for (double g : {pxl_fm_asin(v), pxl_fm_asin<2>(v)})
            if (!std::isnan(g)) { ++special_bad; fprintf(stderr, "asin(%a) = %a, expected NaN\n", v, g); }
    for (double v : {0.0, -0.0, 1.0, -1.0, 0.5, -0.5, -1.5, (double)inf, (double)nan, 5e-324, 1e-200}) {
        double g = pxl_fm_asin(v), w = std::asin(v);
        if (!(same_bits(g, w) || err_ulp(g, asinl((long double)v)) <= 1.0) || (!std::isnan(w) && std::signbit(g) != std::signbit(w))) { ++special_bad; fprintf(stderr, "asin(%g) = %a, libm %a\n", v, g, w); }
    }
    {
        double sn = 7, cs = 7;
        if (pxl_fm_sincos(1e6, &sn, &cs) || pxl_fm_sincos(inf, &sn, &cs) || pxl_fm_sincos(nan, &sn, &cs) || sn != 7 || cs != 7) ++special_bad;
        pxl_fm_sincos(0.0, &sn, &cs);
        if (!(sn == 0.0 && cs == 1.0)) ++special_bad;
    }
    for (double y : sp)
        for (double x : sp)
            if (pxl_fm_atan2_is_tame(y, x) && !same_bits(pxl_fm_atan2<true>(y, x), pxl_fm_atan2(y, x))) ++tame_bad;
    if (pxl_fm_atan2_is_tame(1.0, inf) || pxl_fm_atan2_is_tame(nan, 1.0) || pxl_fm_atan2_is_tame(1.0, nan) || pxl_fm_atan2_is_tame(1.0, -1.0) ||
        pxl_fm_atan2_is_tame(1.0, 0.0) || pxl_fm_atan2_is_tame(inf, 1.0) || pxl_fm_atan2_is_tame(1e-300, 1e-300)) ++tame_bad;
    for (double v : {0.0, -0.0, 0.5, -0.5, 5e-324, 1e-200}) if (!same_bits(pxl_fm_asin<1>(v), pxl_fm_asin(v))) ++half_bad;
    for (double v : {1.0, -1.0, 0.5000000000000001, 1.0000000000000002, -1.5, (double)inf, (double)nan}) if (!same_bits(pxl_fm_asin<2>(v), pxl_fm_asin(v))) ++half_bad;
    if (!(std::isnan(pxl_fm_rsqrt(inf)) && std::isnan(pxl_fm_rsqrt(nan)) && pxl_fm_rsqrt(4.0) == 0.5)) ++special_bad;
    printf("{\"samples\": %ld, \"atan2_max_ulp\": %.3f, \"atan2_worst\": [%.17g, %.17g], \"asin_max_ulp\": %.3f, \"asin_worst\": %.17g, "
           "\"sin_max_ulp\": %.3f, \"sin_worst\": %.17g, \"cos_max_ulp\": %.3f, \"cos_worst\": %.17g, \"sin_max_ulp_big\": %.3f, \"cos_max_ulp_big\": %.3f, \"rsqrt_max_ulp\": %.3f, "
           "\"rsqrt_worst\": %.17g, \"tame_samples\": %ld, \"tame_bad\": %ld, \"asin_half_bad\": %ld, \"special_bad\": %d, \"sincos_near_kpio2_max_abs\": %.4g, \"sincos_near_kpio2_worst\": %.17g}\n",
           n, e_atan2, w_atan2[0], w_atan2[1], e_asin, w_asin, e_sin, w_sin, e_cos, w_cos, e_sin_big, e_cos_big, e_rsqrt, w_rsqrt, n_tame, tame_bad, half_bad, special_bad, e_near_abs, w_near);
    return 0;
}
