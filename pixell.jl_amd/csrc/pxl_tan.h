// pxl_tan.h -- Gnomonic (TAN) evaluators; included by pxl_kernels.hip (one translation unit, -ffp-contract=off).
#pragma once

// ------------------------------------------------------------------------------------------------
// Gnomonic (A16), tan_proj.jl:44-75
// ------------------------------------------------------------------------------------------------
// These are tolerance-checked paths (FP64 transcendentals: pxl_fastmath.h's atan2 / asin / sincos / rsqrt, <= 1.5 ulp, vs glibc), held to the reference's own bar for
// its fast Gnomonic code against wcslib: sum |difference| < 1e-9 over the 1827 x 1825 posmap (test_geometry.jl:116-119).
//
// sky2pix keeps the reference's operations one for one; each angle's sine and cosine come from ONE sincos (one
// argument reduction instead of two; pxl_fastmath.h's for angles below 2^19 pi/2, the library's beyond) and the loop-invariant scale / unit is divided once on the host (a correctly
// rounded quotient either way).
//
// pix2sky is evaluated in an algebraically equal form without the intermediate angles.  The reference
// (tan_proj.jl:59-75) forms D = atan(r), B = atan(-X, Y) and then only ever uses sin D, cos D, sin B, cos B:
//      sin D = r / s,  cos D = 1 / s,  cos B = Y / r,  sin B = -X / r        (r = hypot(X, Y), s = sqrt(1 + r^2))
//      XX = sin d0 sinD cosB + cos d0 cosD = (sin d0 Y + cos d0) / s
//      YY = sinD sinB                      = -X / s
//      alpha = a0 + atan(YY, XX)           = a0 + atan(-X, sin d0 Y + cos d0)          (s > 0 cancels)
//      delta = asin(sin d0 cosD - cos d0 sinD cosB) = asin((sin d0 - cos d0 Y) / s)
// i.e. one rsqrt, one atan2 and one asin instead of sqrt, atan, two atan2, two sin, two cos and asin.  On the
// reference's patch the two forms differ by at most one ulp of the angle, 3e-11 (RA) / 6e-11 (DEC) summed over the
// 3.3 M pixels against the 1e-9 allowed (glibc on both sides; tests/test_gpu_parity.py holds the device to the bound).
struct TanParams { double scale, unit, a0, d0, sd0, cd0, cpx, cpy, su, uos; };
static TanParams tan_setup(const pxl_car_wcs& w) {
    TanParams t;
    t.scale = 1.0 / w.cdelt[0];
    t.unit = w.unit;
    t.a0 = w.crval[0] * (PXL_PI_D / 180);   // deg2rad.(wcs.crval), tan_proj.jl:47
    t.d0 = w.crval[1] * (PXL_PI_D / 180);
    t.sd0 = sin(t.d0); t.cd0 = cos(t.d0);
    t.cpx = w.crpix[0]; t.cpy = w.crpix[1];
    t.su = t.scale / t.unit;                 // the left-to-right head of `scale / unit / (...)`, tan_proj.jl:50
    t.uos = t.unit / t.scale;                // pix2sky: `(crpix - i) * unit / scale` (tan_proj.jl:62-63) as one multiplication -- a
                                             // second rounding of X at most, inside the tolerance these paths are held to
    return t;
}
__device__ inline void tan_sky2pix(const TanParams& t, double a, double d, double* x, double* y) {
    double sd, cd, sa, ca;
    const double da = a - t.a0;
    const bool fast = (int)pxl_fm_sincos(d, &sd, &cd) & (int)pxl_fm_sincos(da, &sa, &ca);    // |angle| <= 2^19 pi/2: <= 1.5 ulp, ~40 instructions each
    if (__builtin_expect(!fast, 0)) {                                                // huge or non-finite angles: the library
        sincos(d, &sd, &cd);
        sincos(da, &sa, &ca);
    }
    double A = cd * ca;
    double F = t.su / (t.sd0 * sd + A * t.cd0);
    double LINE = -F * (t.cd0 * sd - A * t.sd0);
    double SAMPLE = -F * cd * sa;
    *x = t.cpx - SAMPLE;
    *y = t.cpy - LINE;
}
// the row-dependent half of pix2sky: Y and the two combinations of it that every pixel of a row shares
struct TanRow { double Y2, den, num; };
__device__ inline TanRow tan_row(const TanParams& t, double j) {
    const double Y = (t.cpy - j) * t.uos;
    return TanRow{Y * Y, t.sd0 * Y + t.cd0, t.sd0 - t.cd0 * Y};
}
// Domain note: plane coordinates whose squares overflow (|pixel offset| * pixel size beyond 1e150 rad) or that are infinite give
// NaN in DEC here (Inf * 0 inside the reciprocal square root), where the reference's sequence of angles happens to leave a finite
// number; RA keeps the reference's limit.  NaN in, NaN out.  (Routing such points through the reference's own operation order
// inside the kernel was built and dropped: the out-of-line call costs the hot path 24 VGPRs and a stack frame.)
// GRID: the lanes of a wave hold neighbouring pixels of one row (posmap), so their sines fall into one half of asin together
template <bool GRID = false>
__device__ inline void tan_pix2sky_xrow(const TanParams& t, const TanRow& r, double X, double XX, double* a, double* d) {
    const double rs = pxl_fm_rsqrt(1.0 + (XX + r.Y2));
    // in front of the tangent plane's horizon with ordinary magnitudes (every lane of the wave: one vote) atan2 needs none of
    // its infinity / scaling / negative-x / NaN steps -- a sixth of its instructions; same bits either way
    if (__all(pxl_fm_atan2_is_tame(-X, r.den))) *a = t.a0 + pxl_fm_atan2<true>(-X, r.den);
    else                                        *a = t.a0 + pxl_fm_atan2<false>(-X, r.den);
    // likewise asin on a grid: a wave whose sines are all within 1/2 (|dec| <= 30 degrees), or all beyond, evaluates that half
    // only (posmap 0.35-0.38 -> 0.30-0.33 ms).  Scattered points of a wide patch mix the halves in most waves, and two votes cost
    // them 3 %: they take the |v| <= 1/2 vote alone (neutral on a 68-degree patch, 62 -> 69 % on a 4-degree one)
    const double sv = r.num * rs;
    const bool small = fabs(sv) <= 0.5;
    if (__all(small))               *d = pxl_fm_asin<1>(sv);         // also for scattered points: patches within 30 degrees of the equator
    else if (GRID && __all(!small)) *d = pxl_fm_asin<2>(sv);
    else                            *d = pxl_fm_asin<0>(sv);
}
__device__ inline void tan_pix2sky_row(const TanParams& t, const TanRow& r, double i, double* a, double* d) {
    const double X = (t.cpx - i) * t.uos;
    tan_pix2sky_xrow(t, r, X, X * X, a, d);
}
__device__ inline void tan_pix2sky(const TanParams& t, double i, double j, double* a, double* d) {
    tan_pix2sky_row(t, tan_row(t, j), i, a, d);
}

// Two points per lane, 16-byte accesses (VEC: all four arrays 16-byte aligned), one contiguous chunk per block.  (A block that
// takes eight consecutive chunks, with the next chunk's loads issued before the arithmetic of the current one, measured 3-7 %
// slower, A/B in one process: profiles/r03_ab_tan_trips.jsonl.)
template <bool VEC, bool INVERSE>
__global__ __launch_bounds__(256) void k_tan_points(TanParams t, int64_t n, const double* __restrict__ in1,
                                                    const double* __restrict__ in2, double* __restrict__ out1,
                                                    double* __restrict__ out2) {
    const int64_t npair = (n + 1) / 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npair; p += stride) {
        const int64_t k = 2 * p;
        const bool two = k + 1 < n;
        double u[2], v[2], r[2], s[2];
        if (VEC && two) {
            const double2 a = *reinterpret_cast<const double2*>(in1 + k), b = *reinterpret_cast<const double2*>(in2 + k);
            u[0] = a.x; u[1] = a.y; v[0] = b.x; v[1] = b.y;
        } else {
            u[0] = in1[k]; v[0] = in2[k];
            u[1] = two ? in1[k + 1] : u[0]; v[1] = two ? in2[k + 1] : v[0];
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if (INVERSE) tan_pix2sky(t, u[e], v[e], &r[e], &s[e]);      // (i, j) -> (ra, dec)
            else         tan_sky2pix(t, u[e], v[e], &r[e], &s[e]);      // (ra, dec) -> (x, y)
        }
        if (VEC && two) {
            *reinterpret_cast<double2*>(out1 + k) = make_double2(r[0], r[1]);
            *reinterpret_cast<double2*>(out2 + k) = make_double2(s[0], s[1]);
        } else {
            out1[k] = r[0]; out2[k] = s[0];
            if (two) { out1[k + 1] = r[1]; out2[k + 1] = s[1]; }
        }
    }
}

// posmap of a Gnomonic map: a block covers 512 adjacent RA pixels (two per lane, 16-byte stores when the row pitch allows) of
// PXL_TAN_ROWS consecutive rows; the column's share of the arithmetic and the constants are set up once per block (4-7 % faster
// than one row per block, same A/B).
#define PXL_TAN_ROWS 8
template <bool VEC>
__global__ __launch_bounds__(256) void k_posmap_tan(TanParams t, int64_t nx, int64_t row0, int64_t nrows, int64_t nchunk,
                                                    double* __restrict__ ra, double* __restrict__ dec) {
    const int64_t b = blockIdx.x;
    const int64_t jb = (b / nchunk) * PXL_TAN_ROWS;                             // first row of the block (within the request)
    const int64_t i = ((b % nchunk) * blockDim.x + threadIdx.x) * 2;            // 0-based column of the lane's first pixel
    if (i >= nx) return;
    const bool two = i + 1 < nx;
    const double X0 = (t.cpx - (double)(i + 1)) * t.uos, X1 = (t.cpx - (double)(i + 2)) * t.uos;
    const double XX0 = X0 * X0, XX1 = X1 * X1;
#pragma unroll 1
    for (int q = 0; q < PXL_TAN_ROWS; ++q) {
        const int64_t jr = jb + q;
        if (jr >= nrows) break;
        const TanRow r = tan_row(t, (double)(row0 + jr + 1));
        double a[2], d[2];
        tan_pix2sky_xrow<true>(t, r, X0, XX0, &a[0], &d[0]);
        tan_pix2sky_xrow<true>(t, r, X1, XX1, &a[1], &d[1]);
        const int64_t o = jr * nx + i;
        if (VEC && two) {
            *reinterpret_cast<double2*>(ra + o) = make_double2(a[0], a[1]);
            *reinterpret_cast<double2*>(dec + o) = make_double2(d[0], d[1]);
        } else {
            ra[o] = a[0]; dec[o] = d[0];
            if (two) { ra[o + 1] = a[1]; dec[o + 1] = d[1]; }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// posmap of a Gnomonic map, round 4: the trigonometry leaves the pixel loop.
//
// k_posmap_tan above spends ~100 FP64 instructions per pixel (a reciprocal square root, an atan2 and an asin) and is bound by
// their issue, 43 % of the write roofline.  On a GRID the two angles are smooth functions of the column along a row, and the
// angle between two directions of one row has a closed form that is cheap and exact:
//      RA_i  - RA_a  = atan( D (X_a - X_i) / (D^2 + X_i X_a) )                                   (D = sin d0 Y + cos d0, the row's)
//      DEC_i - DEC_a = atan( num (X_a - X_i)(X_a + X_i) / ((rho_a + rho_i)(rho_i rho_a + num^2)) ),   rho = sqrt(X^2 + D^2)
// (tan of a difference; num = sin d0 - cos d0 Y; tan DEC = num / rho, tan(RA - a0) = -X / D).  For pixels of one 128-column tile
// the arguments are below 1/16, so each is a 7-term odd series -- accurate to 1e-19 rad because the DIFFERENCE is small.
// One wave per tile of 128 columns x 64 rows:
//   phase A (a lane = a row): the row's anchor (RA_a, DEC_a) at tile column 64 by the exact evaluator; the differences at nine
//            equispaced node columns and at two check columns by the closed forms; the degree-8 interpolant through the nodes is
//            checked against the closed form at the check columns (tolerance 2^-55 rad); node values go to LDS
//   phase B (a lane = two adjacent columns, marching down the rows): RA = RA_a + sum_k w[col][k] dRA_k, likewise DEC -- 18 FMAs
//            per pixel, with the Lagrange weights of the lane's columns in registers and a row's node values read from LDS
// A row that fails a precondition (D <= 0: beyond the pole, where atan2 has its cut; an argument above 1/16: a tile next to a
// pole or very coarse pixels; a failed check; non-finite values) is evaluated per pixel by tan_pix2sky_xrow, as before.
// Error against the exact per-pixel evaluation: the anchor's own (<= 1.5 ulp, shared by the row's 128 pixels) plus one rounding.
// ------------------------------------------------------------------------------------------------
#define PXL_TG_W 128
#define PXL_TG_ROWS 64
#define PXL_TG_NODES 9
#define PXL_TG_SMAX 0.0625
#define PXL_TG_TOL 0x1p-55
struct TanGridWeights { double w[PXL_TG_W][PXL_TG_NODES]; double chk[2][PXL_TG_NODES]; };
constexpr double tg_lag(double u, int a) {           // Lagrange basis a of the nodes k * (W - 1) / (NODES - 1) at column u
    const double h = (PXL_TG_W - 1.0) / (PXL_TG_NODES - 1);
    double num = 1.0, den = 1.0;
    for (int c = 0; c < PXL_TG_NODES; ++c)
        if (c != a) { num *= (u - c * h); den *= ((a - c) * h); }
    return num / den;
}
#define PXL_TG_CHK0 7.5
#define PXL_TG_CHK1 119.5
constexpr TanGridWeights make_tan_grid_weights() {
    TanGridWeights t{};
    for (int u = 0; u < PXL_TG_W; ++u)
        for (int a = 0; a < PXL_TG_NODES; ++a) t.w[u][a] = tg_lag((double)u, a);
    for (int a = 0; a < PXL_TG_NODES; ++a) { t.chk[0][a] = tg_lag(PXL_TG_CHK0, a); t.chk[1][a] = tg_lag(PXL_TG_CHK1, a); }
    return t;
}
__constant__ TanGridWeights c_tan_grid_weights = make_tan_grid_weights();

// atan(s) for |s| <= 1/16: s (1 - z/3 + z^2/5 - ... + z^6/13), z = s^2; the next term is below 2^-60 s
__device__ inline double tg_atan_small(double s) {
    const double z = s * s;
    double p = 1.0 / 13;
    p = __builtin_fma(p, z, -1.0 / 11);
    p = __builtin_fma(p, z, 1.0 / 9);
    p = __builtin_fma(p, z, -1.0 / 7);
    p = __builtin_fma(p, z, 1.0 / 5);
    p = __builtin_fma(p, z, -1.0 / 3);
    return __builtin_fma(s * z, p, s);
}
// n / d to <= 1 ulp from the reciprocal seed (d finite, non-zero, ordinary magnitude)
__device__ inline double tg_div(double n, double d) {
    double r = pxl_fm_rcp_seed(d);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    const double q = n * r;
    return __builtin_fma(__builtin_fma(-d, q, n), r, q);
}
// the two differences of a row against the anchor at Xa, at the abscissa Xa - dX; *ok: the preconditions of the small-angle forms.
// dX comes from the COLUMN difference, (column - 64) * unit/scale, not from a subtraction of two rounded abscissae: the rounding of
// X itself (half an ulp of ~0.3 rad = 3e-17, i.e. 2e-13 column) would otherwise enter the differences as noise of 1e-16 rad that no
// smooth interpolant can follow -- a first version failed its own check in 87 % of the rows that way.  In the products below a
// rounded X is harmless (a relative 1e-16 of a difference of 1e-2 rad).
__device__ inline void tg_delta(const TanRow& rw, double D2, double N2, double Xa, double rho_a, double dX, double* dra, double* ddec, bool* ok) {
    const double X = Xa - dX;
    const double dot = __builtin_fma(X, Xa, D2);
    const double s = tg_div(rw.den * dX, dot);
    const double rho = __builtin_sqrt(__builtin_fma(X, X, D2));
    const double s2 = tg_div((rw.num * dX) * (Xa + X), (rho_a + rho) * __builtin_fma(rho, rho_a, N2));
    *dra = tg_atan_small(s);
    *ddec = tg_atan_small(s2);
    *ok = dot > 0.0 && __builtin_fabs(s) <= PXL_TG_SMAX && __builtin_fabs(s2) <= PXL_TG_SMAX;      // (NaN fails every comparison)
}

template <bool VEC>
__global__ __launch_bounds__(256) void k_posmap_tan_grid(TanParams t, int64_t nx, int64_t row0, int64_t nrows, int64_t ntx,
                                                         int64_t per, int fronts, double* __restrict__ ra, double* __restrict__ dec) {
    __shared__ __attribute__((aligned(16))) double nodes[4][PXL_TG_ROWS][2 * PXL_TG_NODES + 2];      // per wave and row: 9 dRA, 9 dDEC, RA_a, DEC_a
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // EIGHT WRITE FRONTS, as in k_posmap_car (a write-only stream swept as one front takes 5.3 TB/s, dealt into eight parts of the
    // map 5.9 and more): blocks b and b + 8 share an XCD, so block b works in part b % 8 of the map's tile rows.  `fronts` = 1
    // for small maps (per = all the tile rows)
    const int64_t nbx = (ntx + 3) / 4;                                           // blocks per tile row (4 tiles side by side)
    const int64_t v = fronts > 1 ? (int64_t)(blockIdx.x % fronts) : 0, jb = fronts > 1 ? (int64_t)(blockIdx.x / fronts) : (int64_t)blockIdx.x;
    const int64_t ty = v * per + jb / nbx, tx = (jb % nbx) * 4 + w;
    if (jb / nbx >= per || tx >= ntx) return;                                   // (whole wave; no block-level barrier is used)
    const int64_t ti0 = tx * PXL_TG_W, tj0 = ty * PXL_TG_ROWS;                  // tile origin: 0-based column, row within the request
    if (tj0 >= nrows) return;
    const int nr = (int)((nrows - tj0) < PXL_TG_ROWS ? (nrows - tj0) : PXL_TG_ROWS);
    const double h = (PXL_TG_W - 1.0) / (PXL_TG_NODES - 1);
    // ---- phase A: lane = row
    bool row_ok = false;
    {
        const TanRow rw = tan_row(t, (double)(row0 + tj0 + lane + 1));
        const double Xa = (t.cpx - (double)(ti0 + PXL_TG_W / 2 + 1)) * t.uos;   // anchor: tile column 64
        double ra_a, dec_a;
        tan_pix2sky_xrow<false>(t, rw, Xa, Xa * Xa, &ra_a, &dec_a);
        const double D2 = rw.den * rw.den, N2 = rw.num * rw.num;
        const double rho_a = __builtin_sqrt(__builtin_fma(Xa, Xa, D2));
        bool ok = lane < nr && rw.den > 0.0 && ra_a == ra_a && dec_a == dec_a;
        double dr[PXL_TG_NODES], dd[PXL_TG_NODES];
#pragma unroll
        for (int k = 0; k < PXL_TG_NODES; ++k) {
            bool o;
            tg_delta(rw, D2, N2, Xa, rho_a, (k * h - PXL_TG_W / 2) * t.uos, &dr[k], &dd[k], &o);     // column k h of the tile
            ok = ok && o;
        }
        const double chk_col[2] = {PXL_TG_CHK0, PXL_TG_CHK1};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            double er, ed; bool o;
            tg_delta(rw, D2, N2, Xa, rho_a, (chk_col[q] - PXL_TG_W / 2) * t.uos, &er, &ed, &o);
            double sr = 0.0, sd = 0.0;
#pragma unroll
            for (int k = 0; k < PXL_TG_NODES; ++k) {
                sr = __builtin_fma(c_tan_grid_weights.chk[q][k], dr[k], sr);
                sd = __builtin_fma(c_tan_grid_weights.chk[q][k], dd[k], sd);
            }
            ok = ok && o && __builtin_fabs(sr - er) <= PXL_TG_TOL && __builtin_fabs(sd - ed) <= PXL_TG_TOL;
        }
        row_ok = ok;
        double* nd = nodes[w][lane];
#pragma unroll
        for (int k = 0; k < PXL_TG_NODES; ++k) { nd[k] = dr[k]; nd[PXL_TG_NODES + k] = dd[k]; }
        nd[2 * PXL_TG_NODES] = ra_a; nd[2 * PXL_TG_NODES + 1] = dec_a;
    }
    const unsigned long long okmask = __ballot(row_ok);
    // the LDS writes above are this wave's own: a wave-level wait orders them before its reads below
    __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    // ---- phase B: lane = two adjacent columns
    const int64_t i = ti0 + 2 * lane;
    if (i >= nx) return;
    const bool two = i + 1 < nx;
    double wk[2][PXL_TG_NODES];
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int k = 0; k < PXL_TG_NODES; ++k) wk[e][k] = c_tan_grid_weights.w[2 * lane + e][k];
    const double X0 = (t.cpx - (double)(i + 1)) * t.uos, X1 = (t.cpx - (double)(i + 2)) * t.uos;
#pragma unroll 1
    for (int r = 0; r < nr; ++r) {
        double a[2], d[2];
        if ((okmask >> r) & 1ull) {
            const double* nd = nodes[w][r];
            double s0 = 0.0, s1 = 0.0, u0 = 0.0, u1 = 0.0;
#pragma unroll
            for (int k = 0; k < PXL_TG_NODES; ++k) {
                const double vr = nd[k], vd = nd[PXL_TG_NODES + k];
                s0 = __builtin_fma(wk[0][k], vr, s0); s1 = __builtin_fma(wk[1][k], vr, s1);
                u0 = __builtin_fma(wk[0][k], vd, u0); u1 = __builtin_fma(wk[1][k], vd, u1);
            }
            const double ra_a = nd[2 * PXL_TG_NODES], dec_a = nd[2 * PXL_TG_NODES + 1];
            a[0] = ra_a + s0; a[1] = ra_a + s1; d[0] = dec_a + u0; d[1] = dec_a + u1;
        } else {
            const TanRow rw = tan_row(t, (double)(row0 + tj0 + r + 1));
            tan_pix2sky_xrow<true>(t, rw, X0, X0 * X0, &a[0], &d[0]);
            tan_pix2sky_xrow<true>(t, rw, X1, X1 * X1, &a[1], &d[1]);
        }
        const int64_t o = (tj0 + r) * nx + i;
        if (VEC && two) {
            *reinterpret_cast<double2*>(ra + o) = make_double2(a[0], a[1]);
            *reinterpret_cast<double2*>(dec + o) = make_double2(d[0], d[1]);
        } else {
            ra[o] = a[0]; dec[o] = d[0];
            if (two) { ra[o + 1] = a[1]; dec[o + 1] = d[1]; }
        }
    }
}
