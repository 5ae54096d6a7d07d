/*
 * pixell_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C, single-source restatement of the Pixell.jl (v0.2.9) CAR pixel<->sky hot path,
 * written op-for-op from the reference's Julia source so that the HIP kernels in
 * pixell.jl_amd/csrc/ can be checked bit-for-bit.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library; the product library (libpixell_hip.so)
 * never links, loads or falls back to it.
 *
 * Parity status (see DESIGN.md "Oracle"):
 *   - pix2sky / sky2pix / rewind / geometry / slice_geometry / pixareamap / Gnomonic: PINNED by the
 *     known-answer literals of the reference's own tests (test/test_geometry.jl, test/test_enmap.jl,
 *     docstrings) and the reference's data files test/data/{fullsky,box}_pixareas.dat, transcribed
 *     under tests/golden/, plus wcslib-7.6 vectors mirroring test_geometry.jl:66-80.
 *   - unwind! (DSP.jl unwrap, third-party, not in /root/reference) and the bilinear sampler
 *     (absent from the reference): PARITY UNPINNED -- this file is their definition.
 *
 * Build: gcc -O2 -ffp-contract=off (no FMA contraction: Julia emits separate fmul/fadd for
 * `a0 + (i - i0) * d`, car_proj.jl:104).  All citations are into /root/reference/src/.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct pxl_car_wcs {   /* CarClenshawCurtis{Float64} / CarFejer1{Float64}, projections/car_proj.jl:7-19 */
    double cdelt[2];
    double crpix[2];
    double crval[2];
    double unit;               /* conversion factor to radians (pi/180 for degrees, enmap_geom.jl:18) */
} pxl_car_wcs;

#define PXL_PI     3.141592653589793      /* Float64(pi) */
#define PXL_TWOPI  6.283185307179586      /* 2 * Float64(pi): what Julia's `2pi` literal evaluates to */

enum { PXL_WRAP_NONE = 0, PXL_WRAP_REWIND = 1, PXL_WRAP_UNWIND = 2 };
enum { PXL_FORM_RECIP = 0, PXL_FORM_DIV = 1, PXL_FORM_RECIP_AV = 2 };

static int g_threads = 1;
void pxl_oracle_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int pxl_oracle_get_threads(void) { return g_threads; }
int pxl_oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---- Julia Base.mod(x::Float64, y::Float64) (base/float.jl): r = rem(x,y) (C fmod);
 *      r == 0 -> copysign(r, y); (r > 0) xor (y > 0) -> r + y; else r.                      */
double pxl_jl_mod(double x, double y) {
    double r = fmod(x, y);
    if (r == 0.0) return copysign(r, y);
    if ((r > 0.0) != (y > 0.0)) return r + y;
    return r;
}

/* ---- rewind, enmap_ops.jl:10-19: ref .+ mod.(angles .- ref .+ period/2, period) .- period/2,
 *      evaluated left to right.                                                              */
double pxl_rewind_cpu(double a, double period, double ref) {
    double half = period / 2;
    return (ref + pxl_jl_mod((a - ref) + half, period)) - half;
}

void pxl_rewind_array_cpu(int64_t n, int64_t stride, double* a, double period, double ref) {
    for (int64_t k = 0; k < n; ++k) a[k * stride] = pxl_rewind_cpu(a[k * stride], period, ref);
}

/* ---- unwind!, enmap_ops.jl:26-32, on one strided row of length n:
 *      rewind!; .-= ref; DSP.unwrap!(range=period); .+= ref.
 *      DSP.jl 0.7 unwrap! along one dim is accumulate!((x, y) -> y - round((y - x)/range)*range)
 *      with Julia round = ties-to-even (rint).  PARITY UNPINNED (DSP.jl is not in /root/reference
 *      and no reference test exercises this branch on more than one point).                  */
void pxl_unwind_row_cpu(int64_t n, int64_t stride, double* a, double period, double ref) {
    double prev = 0.0;
    for (int64_t k = 0; k < n; ++k) {
        double m = pxl_rewind_cpu(a[k * stride], period, ref) - ref;
        double y = (k == 0) ? m : m - rint((m - prev) / period) * period;
        prev = y;
        a[k * stride] = y + ref;
    }
}

/* ---- pix2sky! on a 2xN column-major array, car_proj.jl:92-115 (A9).
 *      wrap_mode NONE = safe=false; UNWIND = safe=true (what the reference does for arrays);
 *      REWIND = per-element rewind (what the scalar method does, car_proj.jl:148-150).       */
int pxl_pix2sky_car_f64_cpu(const pxl_car_wcs* w, int64_t n, const double* pix, double* sky, int wrap_mode) {
    double a0 = w->crval[0] * w->unit, d0 = w->crval[1] * w->unit;
    double da = w->cdelt[0] * w->unit, dd = w->cdelt[1] * w->unit;
    double ia0 = w->crpix[0], id0 = w->crpix[1];
    #pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int64_t k = 0; k < n; ++k) {
        double ia = pix[2 * k], id = pix[2 * k + 1];
        double a = a0 + (ia - ia0) * da;
        double d = d0 + (id - id0) * dd;
        if (wrap_mode == PXL_WRAP_REWIND) {
            a = pxl_rewind_cpu(a, PXL_TWOPI, 0.0);
            d = pxl_rewind_cpu(d, PXL_TWOPI, 0.0);
        }
        sky[2 * k] = a;
        sky[2 * k + 1] = d;
    }
    if (wrap_mode == PXL_WRAP_UNWIND) {   /* unwind!(skycoords; dims=2): both rows, period 2pi, ref 0 */
        pxl_unwind_row_cpu(n, 2, sky, PXL_TWOPI, 0.0);
        pxl_unwind_row_cpu(n, 2, sky + 1, PXL_TWOPI, 0.0);
    }
    return 0;
}

/* ---- pix2sky(shape, wcs, ra_pixel, dec_pixel) broadcast over two equal-length vectors,
 *      car_proj.jl:141-152 (A10).  safe=true -> rewind (not unwind; FIXME in the reference).  */
int pxl_pix2sky_car_soa_f64_cpu(const pxl_car_wcs* w, int64_t n, const double* ipix, const double* jpix,
                                double* ra, double* dec, int safe) {
    double a0 = w->crval[0] * w->unit, d0 = w->crval[1] * w->unit;
    double da = w->cdelt[0] * w->unit, dd = w->cdelt[1] * w->unit;
    double ia0 = w->crpix[0], id0 = w->crpix[1];
    #pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int64_t k = 0; k < n; ++k) {
        double a = a0 + (ipix[k] - ia0) * da;
        double d = d0 + (jpix[k] - id0) * dd;
        if (safe) { a = pxl_rewind_cpu(a, PXL_TWOPI, 0.0); d = pxl_rewind_cpu(d, PXL_TWOPI, 0.0); }
        ra[k] = a; dec[k] = d;
    }
    return 0;
}

/* One sky->pix evaluation in each of the reference's three roundings:
 *   RECIP    (A11) car_proj.jl:165-193: ia0 + (a - a0) * (1/da);  period abs(2pi / da)
 *   DIV      (A12) car_proj.jl:220-234: ia0 + (a - a0) / da;      period abs(2pi / da)
 *   RECIP_AV (A13) car_proj.jl:235-252: ia0 + (a - a0) * (1/da);  period abs(2pi * (1/da))
 * center_pix = shape[1:2] ./ 2 .+ 1 (car_proj.jl:186).                                       */
typedef struct { double a0, d0, da, dd, ia0, id0, rda, rdd, cx, cy, px, py; int form, safe; } s2p_t;

static s2p_t s2p_setup(const pxl_car_wcs* w, const int64_t* shape, int safe, int form) {
    s2p_t s;
    s.a0 = w->crval[0] * w->unit; s.d0 = w->crval[1] * w->unit;
    s.da = w->cdelt[0] * w->unit; s.dd = w->cdelt[1] * w->unit;
    s.ia0 = w->crpix[0]; s.id0 = w->crpix[1];
    s.rda = 1 / s.da; s.rdd = 1 / s.dd;
    s.cx = (double)shape[0] / 2 + 1; s.cy = (double)shape[1] / 2 + 1;
    if (form == PXL_FORM_RECIP_AV) { s.px = fabs(PXL_TWOPI * s.rda); s.py = fabs(PXL_TWOPI * s.rdd); }
    else                           { s.px = fabs(PXL_TWOPI / s.da);  s.py = fabs(PXL_TWOPI / s.dd); }
    s.form = form; s.safe = safe;
    return s;
}
static inline void s2p_eval(const s2p_t* s, double a, double d, double* x, double* y) {
    double ix, iy;
    if (s->form == PXL_FORM_DIV) { ix = s->ia0 + (a - s->a0) / s->da;  iy = s->id0 + (d - s->d0) / s->dd; }
    else                         { ix = s->ia0 + (a - s->a0) * s->rda; iy = s->id0 + (d - s->d0) * s->rdd; }
    if (s->safe) { ix = pxl_rewind_cpu(ix, s->px, s->cx); iy = pxl_rewind_cpu(iy, s->py, s->cy); }
    *x = ix; *y = iy;
}

/* ---- sky2pix! on 2xN (A11 by default; `form` selects the rounding) */
int pxl_sky2pix_car_f64_cpu(const pxl_car_wcs* w, const int64_t* shape, int64_t n, const double* sky,
                            double* pix, int safe, int form) {
    s2p_t s = s2p_setup(w, shape, safe, form);
    #pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int64_t k = 0; k < n; ++k) s2p_eval(&s, sky[2 * k], sky[2 * k + 1], &pix[2 * k], &pix[2 * k + 1]);
    return 0;
}

/* ---- sky2pix on two vectors (A13 by default) / scalar broadcast (A12 with form=DIV) */
int pxl_sky2pix_car_soa_f64_cpu(const pxl_car_wcs* w, const int64_t* shape, int64_t n, const double* ra,
                                const double* dec, double* ipix, double* jpix, int safe, int form) {
    s2p_t s = s2p_setup(w, shape, safe, form);
    #pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int64_t k = 0; k < n; ++k) s2p_eval(&s, ra[k], dec[k], &ipix[k], &jpix[k]);
    return 0;
}

/* ---- posmap, enmap_ops.jl:190-203 (A15): per-pixel scalar pix2sky with safe=true (rewind).
 *      Writes rows [row0, row0+nrows) (0-based row0) of the two (nx, ny) maps into ra/dec
 *      buffers of nx*nrows doubles each.                                                     */
int pxl_posmap_car_f64_cpu(const pxl_car_wcs* w, const int64_t* shape, int64_t row0, int64_t nrows,
                           double* ra, double* dec, int safe) {
    double a0 = w->crval[0] * w->unit, d0 = w->crval[1] * w->unit;
    double da = w->cdelt[0] * w->unit, dd = w->cdelt[1] * w->unit;
    double ia0 = w->crpix[0], id0 = w->crpix[1];
    int64_t nx = shape[0];
    #pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int64_t jr = 0; jr < nrows; ++jr) {
        double j = (double)(row0 + jr + 1);
        double d = d0 + (j - id0) * dd;
        if (safe) d = pxl_rewind_cpu(d, PXL_TWOPI, 0.0);
        for (int64_t ii = 0; ii < nx; ++ii) {
            double a = a0 + ((double)(ii + 1) - ia0) * da;
            if (safe) a = pxl_rewind_cpu(a, PXL_TWOPI, 0.0);
            ra[jr * nx + ii] = a;
            dec[jr * nx + ii] = d;
        }
    }
    return 0;
}

/* ---- fullsky_geometry, enmap_geom.jl:47-67 (A4).  res in radians.  shape_io: in = {0,0} to have
 *      it computed (round half-even like Julia round(Int, x)), or a caller-given (nx, ny).
 *      Returns -1 / -2 when the reference's two @asserts (:55-56) would fire.                */
int pxl_fullsky_geometry_cpu(double resx, double resy, int64_t* shape_io, pxl_car_wcs* w) {
    if (shape_io[0] == 0 && shape_io[1] == 0) {
        shape_io[0] = (int64_t)nearbyint(PXL_TWOPI / resx + 0);
        shape_io[1] = (int64_t)nearbyint(PXL_PI / resy + 1);
    }
    int64_t nx = shape_io[0], ny = shape_io[1];
    if (!(fabs(resx * (double)nx - PXL_TWOPI) < 1e-8)) return -1;
    if (!(fabs(resy * (double)(ny - 1) - PXL_PI) < 1e-8)) return -2;
    w->cdelt[0] = -360.0 / (double)nx;
    w->cdelt[1] = 180.0 / (double)(ny - 1);
    w->crpix[0] = floor((double)nx / 2) + 0.5;
    w->crpix[1] = (double)(ny + 1) / 2;
    w->crval[0] = resy * 90 / PXL_PI;
    w->crval[1] = 0.0;
    w->unit = PXL_PI / 180;
    return 0;
}

static double jl_sign(double x) { return x > 0 ? 1.0 : (x < 0 ? -1.0 : x); }
/* Base.rad2deg(z::AbstractFloat) = z * (180 / oftype(z, pi))  (julia base/math.jl; the constant is the
 * Float64 quotient 180/Float64(pi) = 57.29577951308232, formed once, then ONE multiply).  Round 1 restated it
 * as z / (pi/180), which differs by 1 ulp on ~11 % of inputs; tests/test_oracle_reference.py keeps inputs where
 * the two forms differ.  deg2rad(z) = z * (oftype(z, pi) / 180) likewise (tan_proj.jl:47,62).               */
static double jl_rad2deg(double x) { return x * (180 / PXL_PI); }

/* ---- geometry(W, bbox, res), enmap_geom.jl:77-108 (A5).  pos1 = bbox[:,1], pos2 = bbox[:,2],
 *      radians.  Returns -1/-2 for the two divisibility asserts (:82-85).                    */
int pxl_geometry_cpu(const double* pos1, const double* pos2, double resx, double resy,
                     int64_t* shape, pxl_car_wcs* w) {
    if (!(fabs(PXL_TWOPI / resx - nearbyint(PXL_TWOPI / resx)) < 1e-8)) return -1;
    if (!(fabs(PXL_TWOPI / resy - nearbyint(PXL_TWOPI / resy)) < 1e-8)) return -2;
    double res[2] = { resx, resy };
    for (int k = 0; k < 2; ++k) {
        double delta = fabs(pos1[k] - pos2[k]);
        shape[k] = (int64_t)nearbyint(delta / res[k]);
        double mid = (pos1[k] + pos2[k]) / 2;
        double crval = (k == 0) ? mid : 0.0;
        double cdelt = fabs(res[k]) * jl_sign(pos2[k] - pos1[k]);
        w->crpix[k] = 1 - (pos1[k] - crval) / cdelt;
        w->cdelt[k] = jl_rad2deg(cdelt);
        w->crval[k] = jl_rad2deg(crval);
    }
    w->unit = PXL_PI / 180;
    return 0;
}

/* ---- slice_geometry, enmap_ops.jl:154-167 + sliced_wcs car_proj.jl:275-278 (A6).
 *      first/step/last are the Julia range's first(), step(), last() (last already normalised). */
int pxl_slice_geometry_cpu(const pxl_car_wcs* w, const int64_t* first, const int64_t* step, const int64_t* last,
                           int64_t* shape, pxl_car_wcs* out) {
    *out = *w;
    for (int k = 0; k < 2; ++k) {
        if (step[k] == 0) return -1;
        int64_t start = (step[k] > 0) ? first[k] - 1 : first[k];
        int64_t sel_size = last[k] - first[k] + step[k];
        out->crpix[k] = (w->crpix[k] - ((double)start + 0.5)) / (double)step[k] + 0.5;
        out->cdelt[k] = w->cdelt[k] * (double)step[k];
        shape[k] = sel_size / step[k];      /* Julia `div`: truncation toward zero, same as C */
    }
    return 0;
}

/* ---- pixareamap!, enmap_ops.jl:124-138 (N4): one value per row i (1-based), broadcast over RA.
 *      rowarea[i-1] = (sin(d2) - sin(d1)) * abs(cdelt[1]*unit), d from unsafe pix2sky of i -/+ 0.5,
 *      sorted and clamped to [-pi/2, pi/2].                                                  */
int pxl_pixarea_rows_cpu(const pxl_car_wcs* w, int64_t nrows, double* rowarea) {
    double d0 = w->crval[1] * w->unit;
    double da = fabs(w->cdelt[0] * w->unit), dd = w->cdelt[1] * w->unit;
    double id0 = w->crpix[1];
    for (int64_t i = 1; i <= nrows; ++i) {
        double da_ = d0 + (((double)i - 0.5) - id0) * dd;
        double db_ = d0 + (((double)i + 0.5) - id0) * dd;
        double d1 = da_ < db_ ? da_ : db_, d2 = da_ < db_ ? db_ : da_;
        d1 = fmax(-PXL_PI / 2, d1); d2 = fmin(PXL_PI / 2, d2);
        rowarea[i - 1] = (sin(d2) - sin(d1)) * da;
    }
    return 0;
}

/* ---- skyarea_cyl, arbitrary_wcs.jl:125-132 */
double pxl_skyarea_cyl_cpu(const pxl_car_wcs* w, const int64_t* shape) {
    double d0 = w->crval[1] * w->unit, dd = w->cdelt[1] * w->unit, da = w->cdelt[0] * w->unit;
    double id0 = w->crpix[1];
    double da_ = d0 + (0.5 - id0) * dd;
    double db_ = d0 + (((double)shape[1] + 0.5) - id0) * dd;
    double d1 = da_ < db_ ? da_ : db_, d2 = da_ < db_ ? db_ : da_;
    d1 = fmax(-PXL_PI / 2, d1); d2 = fmin(PXL_PI / 2, d2);
    return (sin(d2) - sin(d1)) * fabs(da) * (double)shape[0];
}

/* ---- Gnomonic (TAN) evaluators, projections/tan_proj.jl:44-75 (A16).  `safe` is ignored there. */
int pxl_sky2pix_tan_f64_cpu(const pxl_car_wcs* w, int64_t n, const double* ra, const double* dec,
                            double* x, double* y) {
    double scale = 1.0 / w->cdelt[0];
    double unit = w->unit;
    double a0 = w->crval[0] * (PXL_PI / 180), d0 = w->crval[1] * (PXL_PI / 180);   /* deg2rad.(crval) */
    #pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int64_t k = 0; k < n; ++k) {
        double a = ra[k], d = dec[k];
        double A = cos(d) * cos(a - a0);
        double F = scale / unit / (sin(d0) * sin(d) + A * cos(d0));
        double LINE = -F * (cos(d0) * sin(d) - A * sin(d0));
        double SAMPLE = -F * cos(d) * sin(a - a0);
        x[k] = w->crpix[0] - SAMPLE;
        y[k] = w->crpix[1] - LINE;
    }
    return 0;
}
int pxl_pix2sky_tan_f64_cpu(const pxl_car_wcs* w, int64_t n, const double* ipix, const double* jpix,
                            double* ra, double* dec) {
    double scale = 1.0 / w->cdelt[0];
    double unit = w->unit;
    double a0 = w->crval[0] * (PXL_PI / 180), d0 = w->crval[1] * (PXL_PI / 180);
    #pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int64_t k = 0; k < n; ++k) {
        double X = (w->crpix[0] - ipix[k]) * unit / scale;
        double Y = (w->crpix[1] - jpix[k]) * unit / scale;
        double D = atan(sqrt(X * X + Y * Y));
        double B = atan2(-X, Y);
        double XX = sin(d0) * sin(D) * cos(B) + cos(d0) * cos(D);
        double YY = sin(D) * sin(B);
        ra[k] = a0 + atan2(YY, XX);
        dec[k] = asin(sin(d0) * cos(D) - cos(d0) * sin(D) * cos(B));
    }
    return 0;
}

/* ======================================================================================
 * R1 -- bilinear sample / reproject.  ABSENT FROM THE REFERENCE; this is the definition
 * (SURVEY 8(a) row R1).  PARITY UNPINNED.
 *
 *   (x, y) 1-based Float64 pixel coordinates in the source map (nx, ny[, nc]);
 *   i0 = floor(x), fx = x - i0, j0 = floor(y), fy = y - j0;
 *   v = (1-fy)*((1-fx)*m[i0,j0] + fx*m[i1,j0]) + fy*((1-fx)*m[i0,j1] + fx*m[i1,j1])
 *   column taps wrap modulo nx iff the map spans the full circle
 *       (abs(nx*abs(cdelt[1]*unit) - 2pi) < 1e-8, the test of enmap_geom.jl:55),
 *   otherwise taps outside [1, nx] read as 0; row taps outside [1, ny] read as 0;
 *   rows inside [1, ny] but outside the resident window [row0+1, row0+nrows] also read as 0
 *   (the sharded caller guarantees they are never needed);
 *   a non-finite x or y gives NaN.
 * ====================================================================================== */
int pxl_car_is_periodic_cpu(const pxl_car_wcs* w, int64_t nx) {
    return fabs((double)nx * fabs(w->cdelt[0] * w->unit) - PXL_TWOPI) < 1e-8;
}

typedef struct { const double* src; int64_t nx, ny, row0, nrows; int periodic; } srcmap_t;

static inline double tap(const srcmap_t* m, int64_t i, int64_t j) {   /* i, j 1-based */
    if (j < 1 || j > m->ny) return 0.0;
    int64_t jr = j - 1 - m->row0;
    if (jr < 0 || jr >= m->nrows) return 0.0;
    if (m->periodic) { i = (i - 1) % m->nx; if (i < 0) i += m->nx; i += 1; }
    else if (i < 1 || i > m->nx) return 0.0;
    return m->src[jr * m->nx + (i - 1)];
}

static inline double bilerp(const srcmap_t* m, double x, double y) {
    if (!(isfinite(x) && isfinite(y))) return NAN;
    double fi = floor(x), fj = floor(y);
    double fx = x - fi, fy = y - fj;
    int64_t i0 = (int64_t)fi, j0 = (int64_t)fj;
    double top = (1 - fx) * tap(m, i0, j0) + fx * tap(m, i0 + 1, j0);
    double bot = (1 - fx) * tap(m, i0, j0 + 1) + fx * tap(m, i0 + 1, j0 + 1);
    return (1 - fy) * top + fy * bot;
}

/* Scattered sample: (x,y) = sky2pix!(shape_in, wcs_in, sky2xN; safe=true) [A11], then bilerp.
 * src is (nx, nrows, nc) resident rows [row0, row0+nrows) of each component plane;
 * out is (n, nc) column-major.                                                              */
int pxl_sample_car_bilinear_f64_cpu(const pxl_car_wcs* win, const int64_t* shape_in /*nx,ny,nc*/,
                                    const double* src, int64_t src_row0, int64_t src_nrows,
                                    int64_t n, const double* sky, double* out) {
    s2p_t s = s2p_setup(win, shape_in, 1, PXL_FORM_RECIP);
    int64_t nx = shape_in[0], ny = shape_in[1], nc = shape_in[2];
    int periodic = pxl_car_is_periodic_cpu(win, nx);
    #pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int64_t k = 0; k < n; ++k) {
        double x, y;
        s2p_eval(&s, sky[2 * k], sky[2 * k + 1], &x, &y);
        for (int64_t c = 0; c < nc; ++c) {
            srcmap_t m = { src + c * nx * src_nrows, nx, ny, src_row0, src_nrows, periodic };
            out[c * n + k] = bilerp(&m, x, y);
        }
    }
    return 0;
}

/* Regular-grid separable coordinate tables:
 *   xs[i-1] = x of output column i, ys[j-1] = y of output row j, with
 *   (a, d) = pix2sky(shape_out, wcs_out, i, j; safe=false) [A10] and
 *   (x, y) = sky2pix(shape_in, wcs_in, a, d; safe=true)    [A12, division form].             */
int pxl_reproject_tables_cpu(const pxl_car_wcs* win, const int64_t* shape_in,
                             const pxl_car_wcs* wout, const int64_t* shape_out,
                             double* xs, double* ys) {
    s2p_t s = s2p_setup(win, shape_in, 1, PXL_FORM_DIV);
    double a0 = wout->crval[0] * wout->unit, d0 = wout->crval[1] * wout->unit;
    double da = wout->cdelt[0] * wout->unit, dd = wout->cdelt[1] * wout->unit;
    double ia0 = wout->crpix[0], id0 = wout->crpix[1];
    double dummy;
    for (int64_t i = 1; i <= shape_out[0]; ++i) {
        double a = a0 + ((double)i - ia0) * da;
        s2p_eval(&s, a, s.d0, &xs[i - 1], &dummy);
    }
    for (int64_t j = 1; j <= shape_out[1]; ++j) {
        double d = d0 + ((double)j - id0) * dd;
        s2p_eval(&s, s.a0, d, &dummy, &ys[j - 1]);
    }
    return 0;
}

/* reproject: dst rows [dst_row0, dst_row0+dst_nrows) of the (nx_o, ny_o[, nc]) output map, from
 * source rows [src_row0, src_row0+src_nrows).  dst is (nx_o, dst_nrows, nc) column-major.      */
int pxl_reproject_car_bilinear_f64_cpu(const pxl_car_wcs* win, const int64_t* shape_in /*nx,ny,nc*/,
                                       const double* src, int64_t src_row0, int64_t src_nrows,
                                       const pxl_car_wcs* wout, const int64_t* shape_out /*nx,ny*/,
                                       double* dst, int64_t dst_row0, int64_t dst_nrows) {
    int64_t nx = shape_in[0], ny = shape_in[1], nc = shape_in[2];
    int64_t nxo = shape_out[0], nyo = shape_out[1];
    if (dst_row0 < 0 || dst_row0 + dst_nrows > nyo) return -22;
    double* xs = (double*)malloc(sizeof(double) * (size_t)(nxo + nyo));
    if (!xs) return -12;
    double* ys = xs + nxo;
    pxl_reproject_tables_cpu(win, shape_in, wout, shape_out, xs, ys);
    int periodic = pxl_car_is_periodic_cpu(win, nx);
    for (int64_t c = 0; c < nc; ++c) {
        srcmap_t m = { src + c * nx * src_nrows, nx, ny, src_row0, src_nrows, periodic };
        double* d = dst + c * nxo * dst_nrows;
        #pragma omp parallel for num_threads(g_threads) schedule(static)
        for (int64_t jr = 0; jr < dst_nrows; ++jr) {
            double y = ys[dst_row0 + jr];
            for (int64_t i = 0; i < nxo; ++i) d[jr * nxo + i] = bilerp(&m, xs[i], y);
        }
    }
    free(xs);
    return 0;
}

/* Source rows (0-based, half-open [lo, hi)) that reprojecting output rows
 * [dst_row0, dst_row0+dst_nrows) touches with a possibly non-zero weight or not -- i.e. both
 * j0 and j0+1 of every output row, clipped to the map.  Used by the dec-strip sharding tests.  */
int pxl_reproject_src_rows_cpu(const pxl_car_wcs* win, const int64_t* shape_in,
                               const pxl_car_wcs* wout, const int64_t* shape_out,
                               int64_t dst_row0, int64_t dst_nrows, int64_t* lo, int64_t* hi) {
    int64_t nxo = shape_out[0], nyo = shape_out[1], ny = shape_in[1];
    double* xs = (double*)malloc(sizeof(double) * (size_t)(nxo + nyo));
    if (!xs) return -12;
    double* ys = xs + nxo;
    pxl_reproject_tables_cpu(win, shape_in, wout, shape_out, xs, ys);
    int64_t l = INT64_MAX, h = INT64_MIN;
    for (int64_t jr = 0; jr < dst_nrows; ++jr) {
        double y = ys[dst_row0 + jr];
        if (!isfinite(y)) continue;
        int64_t j0 = (int64_t)floor(y);
        for (int64_t j = j0; j <= j0 + 1; ++j)
            if (j >= 1 && j <= ny) { if (j - 1 < l) l = j - 1; if (j > h) h = j; }
    }
    if (l > h) { l = 0; h = 0; }
    *lo = l; *hi = h;
    free(xs);
    return 0;
}

/* ---- Generic (non-separable) bilinear reprojection between CAR (proj 0) and Gnomonic (proj 1) maps (N2):
 *      per output pixel pix2sky(out) [car_proj.jl:146-147 safe=false | tan_proj.jl:59-75], then
 *      sky2pix(in) [car_proj.jl:225-231 safe=true | tan_proj.jl:44-57], then the R1 2x2 gather.
 *      A sky point behind a Gnomonic source's tangent plane (cos c <= 0) reads as 0.  PARITY UNPINNED
 *      (no reprojection in the reference); libm-level tolerance against the device.                     */
int pxl_reproject_generic_bilinear_f64_cpu(const pxl_car_wcs* win, int proj_in, const int64_t* shape_in,
                                           const double* src, const pxl_car_wcs* wout, int proj_out,
                                           const int64_t* shape_out, double* dst) {
    int64_t nx = shape_in[0], ny = shape_in[1], nc = shape_in[2];
    int64_t nxo = shape_out[0], nyo = shape_out[1];
    int periodic = (proj_in == 0) && pxl_car_is_periodic_cpu(win, nx);
    s2p_t s = s2p_setup(win, shape_in, 1, PXL_FORM_DIV);
    double oa0 = wout->crval[0] * wout->unit, od0 = wout->crval[1] * wout->unit;
    double oda = wout->cdelt[0] * wout->unit, odd = wout->cdelt[1] * wout->unit;
    double ia0 = win->crval[0] * (PXL_PI / 180), id0 = win->crval[1] * (PXL_PI / 180);
    #pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int64_t jr = 0; jr < nyo; ++jr) {
        for (int64_t i = 0; i < nxo; ++i) {
            double ip = (double)(i + 1), jp = (double)(jr + 1), ra, dec, x, y;
            if (proj_out == 1) pxl_pix2sky_tan_f64_cpu(wout, 1, &ip, &jp, &ra, &dec);
            else { ra = oa0 + (ip - wout->crpix[0]) * oda; dec = od0 + (jp - wout->crpix[1]) * odd; }
            int visible = 1;
            if (proj_in == 1) {
                pxl_sky2pix_tan_f64_cpu(win, 1, &ra, &dec, &x, &y);
                visible = (sin(id0) * sin(dec) + cos(dec) * cos(ra - ia0) * cos(id0)) > 0.0;
            } else s2p_eval(&s, ra, dec, &x, &y);
            for (int64_t c = 0; c < nc; ++c) {
                srcmap_t m = { src + c * nx * ny, nx, ny, 0, ny, periodic };
                double v = bilerp(&m, x, y);                 /* NaN when x or y is not finite */
                if (!visible && !isnan(v)) v = 0.0;
                dst[c * nxo * nyo + jr * nxo + i] = v;
            }
        }
    }
    return 0;
}
