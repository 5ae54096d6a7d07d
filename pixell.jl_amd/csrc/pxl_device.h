// pxl_device.h -- device-side arithmetic shared by every kernel of libpixell_hip.so.
//
// Each helper restates one piece of the reference's Float64 arithmetic (file:line under
// /root/reference/src/).  The translation unit is compiled with -ffp-contract=off, so every `a + b * c`
// below is a separate v_mul_f64 / v_add_f64, exactly like the fmul/fadd Julia emits.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pixell_hip.h"

#define PXL_PI_D     3.141592653589793    // Float64(pi)
#define PXL_TWOPI_D  6.283185307179586    // Julia's `2pi` == 2 * Float64(pi)

namespace pxl {

// Exact C fmod(x, y) for y > 0 without the library's bit-serial reduction loop (~100 instructions per call).
// With q an integer within one of trunc(|x|/y), v = |x| - q*y is an integer multiple of ulp(y) below 2y in
// magnitude, and fma evaluates it with a single rounding: whenever the result lands in [0, y) it IS v (every
// multiple of ulp(y) in that range is representable), i.e. the true remainder; otherwise its sign / size says
// which neighbour of q is the quotient, and the fma is redone with it.  The fma here is exact-arithmetic
// machinery, not contraction: the value returned is fmod's, bit for bit (sign of x, also on zero).  Huge
// quotients, NaN/Inf and extreme periods take the library fmod.  ry ~ 1/y (any few-ulp approximation).
// Branch-free attempt: *ok says whether the value returned is fmod(x, y) (it is not for NaN/Inf, quotients
// beyond 2^50 and extreme periods -- the caller then takes the library fmod).
__host__ __device__ inline double fmod_pos_try(double x, double y, double ry, bool* ok) {
    const double ax = fabs(x);
    double q = trunc(ax * ry);
    double r = __builtin_fma(-q, y, ax);
    q = (q - (r < 0.0 ? 1.0 : 0.0)) + (r >= y ? 1.0 : 0.0);        // the neighbour (0 / 1 flags: one v_cndmask each, not two per 64-bit select)
    r = __builtin_fma(-q, y, ax);                                   // exact once q is the true quotient
    *ok = ax < y * 0x1p50 && r >= 0.0 && r < y && y >= 0x1p-900 && y <= 0x1p900;
    return copysign(r, x);
}
__host__ __device__ inline double fmod_pos(double x, double y, double ry) {
    bool ok;
    const double r = fmod_pos_try(x, y, ry, &ok);
    if (__builtin_expect(ok, 1)) return r;
    return fmod(x, y);
}

// Julia Base.mod(x::Float64, y::Float64): r = rem(x, y); r == 0 -> copysign(r, y);
// (r > 0) xor (y > 0) -> r + y; else r.   fmod is exact, so this is bit-identical on any IEEE target.
__host__ __device__ inline double jl_mod(double x, double y, double ry) {
    double r = (y > 0.0) ? fmod_pos(x, y, ry) : fmod(x, y);
    if (r == 0.0) return copysign(r, y);
    if ((r > 0.0) != (y > 0.0)) return r + y;
    return r;
}
__host__ __device__ inline double jl_mod(double x, double y) { return jl_mod(x, y, 1.0 / y); }

// rewind, enmap_ops.jl:10-13: ref + mod(a - ref + period/2, period) - period/2, left to right.
// rperiod ~ 1/period only seeds the quotient guess of fmod_pos; it does not enter the result.
__host__ __device__ inline double rewind(double a, double period, double ref, double rperiod) {
    double half = period / 2;
    return (ref + jl_mod((a - ref) + half, period, rperiod)) - half;
}
__host__ __device__ inline double rewind(double a, double period, double ref) {
    return rewind(a, period, ref, 1.0 / period);
}
// The same for period > 0 without a branch; *ok false -> the caller must use rewind() instead.
__host__ __device__ inline double rewind_try(double a, double period, double ref, double rperiod, bool* ok) {
    const double half = period / 2;
    const double r = fmod_pos_try((a - ref) + half, period, rperiod, ok);
    // jl_mod for y > 0: r + period for r < 0, +0 for r = +-0, r otherwise -- as one fma with a 0 / 1 flag (flag * period is exact,
    // the sum is rounded once like the reference's r + period, and 0 * period + (-0) = +0)
    const double mod = __builtin_fma(r < 0.0 ? 1.0 : 0.0, period, r);
    return (ref + mod) - half;
}
// ref = 0: `0 + mod` is mod itself (mod is never -0, see above), so the addition is dropped
__host__ __device__ inline double rewind0_try(double a, double period, double rperiod, bool* ok) {
    const double half = period / 2;
    const double r = fmod_pos_try((a - 0.0) + half, period, rperiod, ok);
    return __builtin_fma(r < 0.0 ? 1.0 : 0.0, period, r) - half;
}

// Pre-multiplied WCS scalars: the prologues of car_proj.jl:95-98 and :168-173.
struct CarAffine {
    double a0, d0;     // crval .* unit
    double da, dd;     // cdelt .* unit
    double ia0, id0;   // crpix
};
__host__ __device__ inline CarAffine car_affine(const pxl_car_wcs& w) {
    CarAffine c;
    c.a0 = w.crval[0] * w.unit; c.d0 = w.crval[1] * w.unit;
    c.da = w.cdelt[0] * w.unit; c.dd = w.cdelt[1] * w.unit;
    c.ia0 = w.crpix[0]; c.id0 = w.crpix[1];
    return c;
}

// pix -> sky, car_proj.jl:104-105 / :146-147
__host__ __device__ inline double p2s_ra(const CarAffine& c, double i)  { return c.a0 + (i - c.ia0) * c.da; }
__host__ __device__ inline double p2s_dec(const CarAffine& c, double j) { return c.d0 + (j - c.id0) * c.dd; }

// sky -> pix in the reference's three roundings (see PXL_FORM_* in pixell_hip.h)
struct Sky2Pix {
    CarAffine c;
    double rda, rdd;   // 1/da, 1/dd            car_proj.jl:173
    double cx, cy;     // shape[1:2] ./ 2 .+ 1   car_proj.jl:186
    double px, py;     // pixel periods          car_proj.jl:187 / :229-230 / :247-248
    double rpx, rpy;   // ~1/px, ~1/py: quotient seeds for fmod_pos only
    int form, safe;
};
__host__ __device__ inline Sky2Pix sky2pix_setup(const pxl_car_wcs& w, int64_t nx, int64_t ny, int safe, int form) {
    Sky2Pix s;
    s.c = car_affine(w);
    s.rda = 1 / s.c.da; s.rdd = 1 / s.c.dd;
    s.cx = (double)nx / 2 + 1; s.cy = (double)ny / 2 + 1;
    if (form == PXL_FORM_RECIP_AV) { s.px = fabs(PXL_TWOPI_D * s.rda); s.py = fabs(PXL_TWOPI_D * s.rdd); }
    else                           { s.px = fabs(PXL_TWOPI_D / s.c.da); s.py = fabs(PXL_TWOPI_D / s.c.dd); }
    s.rpx = 1.0 / s.px; s.rpy = 1.0 / s.py;
    s.form = form; s.safe = safe;
    return s;
}
__host__ __device__ inline double s2p_x(const Sky2Pix& s, double a) {
    double ix = (s.form == PXL_FORM_DIV) ? s.c.ia0 + (a - s.c.a0) / s.c.da : s.c.ia0 + (a - s.c.a0) * s.rda;
    return s.safe ? rewind(ix, s.px, s.cx, s.rpx) : ix;
}
__host__ __device__ inline double s2p_y(const Sky2Pix& s, double d) {
    double iy = (s.form == PXL_FORM_DIV) ? s.c.id0 + (d - s.c.d0) / s.c.dd : s.c.id0 + (d - s.c.d0) * s.rdd;
    return s.safe ? rewind(iy, s.py, s.cy, s.rpy) : iy;
}

// Split a 1-based Float64 pixel coordinate into integer cell + fraction.  The cell index is clamped to
// +-2^30 so it fits an int32 table entry; anything that far out reads as zero taps anyway.
#define PXL_CELL_LIMIT 1073741824.0
__host__ __device__ inline void split_cell(double x, int32_t* cell, double* frac) {
    double f = floor(x);
    *frac = x - f;
    f = fmin(fmax(f, -PXL_CELL_LIMIT), PXL_CELL_LIMIT);
    *cell = (int32_t)f;
}

// Source map view used by the direct-gather paths (oracle: tap()/bilerp() in oracle/pixell_oracle.c).
// T is the STORAGE type of the map (double, or float for Float32 maps); arithmetic is always Float64.
template <typename T>
struct SrcViewT {
    const T* plane;        // first resident row of this component plane
    int64_t nx, ny;        // full map size
    int64_t row0, nrows;   // resident rows [row0, row0 + nrows), 0-based
    int periodic;          // RA taps wrap modulo nx
};
using SrcView = SrcViewT<double>;
// 1-based column of a periodic map: ((i - 1) mod nx) + 1.  Within one period either side it is a compare and an add
// (safe sky2pix keeps x within half a period of the map centre); the 64-bit modulo (~100 instructions) only beyond.
__device__ inline int64_t wrap_col(int64_t i, int64_t nx) {
    if (i >= 1 - nx && i <= 2 * nx) { if (i > nx) i -= nx; else if (i < 1) i += nx; return i; }
    i = (i - 1) % nx; if (i < 0) i += nx; return i + 1;
}
template <typename T>
__device__ inline double tap(const SrcViewT<T>& m, int64_t i, int64_t j) {   // i, j 1-based
    if (j < 1 || j > m.ny) return 0.0;
    int64_t jr = j - 1 - m.row0;
    if (jr < 0 || jr >= m.nrows) return 0.0;
    if (m.periodic) i = wrap_col(i, m.nx);
    else if (i < 1 || i > m.nx) return 0.0;
    return (double)m.plane[jr * m.nx + (i - 1)];
}
template <typename T>
__device__ inline double bilerp_cells(const SrcViewT<T>& m, int64_t i0, double fx, int64_t j0, double fy) {
    double top = (1 - fx) * tap(m, i0, j0) + fx * tap(m, i0 + 1, j0);
    double bot = (1 - fx) * tap(m, i0, j0 + 1) + fx * tap(m, i0 + 1, j0 + 1);
    return (1 - fy) * top + fy * bot;
}

}  // namespace pxl
