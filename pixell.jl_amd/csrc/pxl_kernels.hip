// pxl_kernels.hip -- gfx950 kernels + C ABI of libpixell_hip.so (see include/pixell_hip.h).
//
// Every kernel here is HBM-bound (streams or gathers of Float64); none is GEMM-shaped, so there is
// no MFMA.  The design rules are the memory ones: 16 B per lane coalesced loads/stores, source rows
// staged through LDS once per output tile, XCD-aware tile order so neighbouring tiles share an L2.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared   (see build.py)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>

#include "pxl_device.h"

using namespace pxl;

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail(PXL_EHIP, "%s: %s", #expr, hipGetErrorString(e_));  \
    } while (0)

static int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(PXL_EHIP, "launch of %s failed: %s", what, hipGetErrorString(e));
    return PXL_OK;
}

static bool wcs_ok(const pxl_car_wcs* w) {
    if (!w) return false;
    for (int k = 0; k < 2; ++k)
        if (!std::isfinite(w->cdelt[k]) || !std::isfinite(w->crpix[k]) || !std::isfinite(w->crval[k]) ||
            w->cdelt[k] == 0.0)
            return false;
    return std::isfinite(w->unit) && w->unit != 0.0;
}

// Grid for 1-D streaming kernels: one contiguous chunk per block (measured best on MI355X: 69-73 % of HBM peak
// vs 57-67 % with a few thousand grid-striding blocks); grid-stride only past 2^20 blocks.
static inline unsigned stream_grid(int64_t work_items, int block) {
    static const int64_t cap = [] { const char* v = getenv("PXL_STREAM_BLOCKS"); return (v && *v) ? atoll(v) : (1LL << 20); }();
    int64_t nb = (work_items + block - 1) / block;
    if (nb < 1) nb = 1;
    if (nb > cap) nb = cap;
    return (unsigned)nb;
}

// ------------------------------------------------------------------------------------------------
// elementwise evaluators (A9-A13): one (c1, c2) pair = 16 B in, 16 B out per lane
// ------------------------------------------------------------------------------------------------
// Each lane handles UNR points per trip, all loads issued before the arithmetic so several 16-B requests
// per lane are in flight (a single dependent load/store per trip left the stream at 57 % of HBM peak).
#define PXL_UNR 4
__global__ __launch_bounds__(256) void k_pix2sky_pairs(CarAffine c, int64_t n, const double2* pix,
                                                       double2* sky, int mode) {
    // mode 0: affine only; 1: rewind; 2: rewind and leave m = rewound - ref for the unwrap passes (ref = 0)
    // a block sweeps contiguous chunks of 256*UNR points (like a copy kernel): the UNR requests of a lane
    // are 4 KiB apart, not a power-of-two number of MiB apart (which camps on one HBM channel)
    const int64_t chunk = (int64_t)blockDim.x * PXL_UNR;
    for (int64_t k0 = (int64_t)blockIdx.x * chunk + threadIdx.x; k0 < n; k0 += (int64_t)gridDim.x * chunk) {
        double2 p[PXL_UNR];
#pragma unroll
        for (int u = 0; u < PXL_UNR; ++u) {
            int64_t k = k0 + u * blockDim.x;
            p[u] = (k < n) ? pix[k] : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int u = 0; u < PXL_UNR; ++u) {
            int64_t k = k0 + u * blockDim.x;
            double a = p2s_ra(c, p[u].x);
            double d = p2s_dec(c, p[u].y);
            if (mode) { a = rewind(a, PXL_TWOPI_D, 0.0); d = rewind(d, PXL_TWOPI_D, 0.0); }
            if (mode == 2) { a = a - 0.0; d = d - 0.0; }        // angles .-= ref_angle  (enmap_ops.jl:28)
            if (k < n) sky[k] = make_double2(a, d);
        }
    }
}

__global__ __launch_bounds__(256) void k_pix2sky_soa(CarAffine c, int64_t n, const double* __restrict__ ip,
                                                     const double* __restrict__ jp, double* __restrict__ ra,
                                                     double* __restrict__ dec, int safe) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) {
        double a = p2s_ra(c, ip[k]);
        double d = p2s_dec(c, jp[k]);
        if (safe) { a = rewind(a, PXL_TWOPI_D, 0.0); d = rewind(d, PXL_TWOPI_D, 0.0); }
        ra[k] = a; dec[k] = d;
    }
}

__global__ __launch_bounds__(256) void k_sky2pix_pairs(Sky2Pix s, int64_t n, const double2* sky,
                                                       double2* pix) {
    const int64_t chunk = (int64_t)blockDim.x * PXL_UNR;
    for (int64_t k0 = (int64_t)blockIdx.x * chunk + threadIdx.x; k0 < n; k0 += (int64_t)gridDim.x * chunk) {
        double2 v[PXL_UNR];
#pragma unroll
        for (int u = 0; u < PXL_UNR; ++u) {
            int64_t k = k0 + u * blockDim.x;
            v[u] = (k < n) ? sky[k] : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int u = 0; u < PXL_UNR; ++u) {
            int64_t k = k0 + u * blockDim.x;
            if (k < n) pix[k] = make_double2(s2p_x(s, v[u].x), s2p_y(s, v[u].y));
        }
    }
}

__global__ __launch_bounds__(256) void k_sky2pix_soa(Sky2Pix s, int64_t n, const double* __restrict__ ra,
                                                     const double* __restrict__ dec, double* __restrict__ ip,
                                                     double* __restrict__ jp) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) {
        ip[k] = s2p_x(s, ra[k]);
        jp[k] = s2p_y(s, dec[k]);
    }
}

// ------------------------------------------------------------------------------------------------
// unwind! (A8, car_proj.jl:110-112 -> enmap_ops.jl:26-32): rewind, subtract ref, DSP.unwrap along the
// point axis, add ref.  With m[k] the rewound value,
//     y[0] = m[0];   y[k] = m[k] - r_k * P,   r_k = rint((m[k] - y[k-1]) / P)        (DSP.jl unwrap kernel)
// y[k-1] is itself m[k-1] - r_{k-1} * P, so the only state the recurrence carries is the INTEGER r_{k-1}:
//     r_k = F_k(r_{k-1}),   F_k(r) = rint((m[k] - (m[k-1] - r*P)) / P)  =  r + c_k   with c_k in {-1,0,1},
// where c_k can depend on r only when (m[k]-m[k-1])/P sits within rounding of a tie.  That makes the scan
// parallel AND exact:
//   1. c_k := rint((m[k] - m[k-1]) / P)                                  (nominal increments, int8)
//   2. r := inclusive prefix sum of c                                    (two-level block scan, int32)
//   3. verify every k with the reference's own floating-point formula: t = rint((m[k] - (m[k-1] - r[k-1]*P))/P);
//      where t != r[k], fix c_k += t - r[k] and raise a flag
//   4. if anything was fixed, repeat 2-3 once (device-gated); a clean verification means r is -- by induction from r_0 = 0 -- exactly
//      the sequential result, and y[k] = m[k] - r[k]*P + ref is written.  Otherwise (adversarial ties) the
//      exact serial kernel below runs instead.  Nothing synchronises with the host.
// PARITY UNPINNED (DSP.jl is not in the reference tree); the oracle's pxl_unwind_row_cpu is the definition.
// ------------------------------------------------------------------------------------------------
#define PXL_SCAN_ITEMS 16
#define PXL_SCAN_BLOCK (256 * PXL_SCAN_ITEMS)

__global__ __launch_bounds__(256) void k_unwrap_incr(int64_t n, int nrow, const double* __restrict__ m2, double period,
                                                     int8_t* __restrict__ c) {
    // m2: nrow x N interleaved rewound values (nrow = 2 for coordinate batches, 1 for a plain vector); c: [nrow][n]
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) {
        for (int row = 0; row < nrow; ++row) {
            int v = 0;
            if (k > 0) v = (int)rint((m2[nrow * k + row] - m2[nrow * (k - 1) + row]) / period);
            c[row * n + k] = (int8_t)v;
        }
    }
}

// local inclusive scan of 4096-element blocks; blockIdx.y = coordinate row
template <typename TIN>
__global__ __launch_bounds__(256) void k_scan_local(int64_t n, const TIN* __restrict__ c, int32_t* __restrict__ rloc,
                                                    int32_t* __restrict__ bsum, int64_t nb, const int32_t* __restrict__ gate) {
    __shared__ int32_t wsum[4];
    if (gate && *gate == 0) return;         // second pass: only if the first verification found mismatches
    const int row = blockIdx.y;
    const int64_t base = (int64_t)blockIdx.x * PXL_SCAN_BLOCK + (int64_t)threadIdx.x * PXL_SCAN_ITEMS;
    int32_t v[PXL_SCAN_ITEMS];
    int32_t run = 0;
#pragma unroll
    for (int i = 0; i < PXL_SCAN_ITEMS; ++i) {
        int64_t k = base + i;
        run += (k < n) ? (int32_t)c[row * n + k] : 0;
        v[i] = run;
    }
    // exclusive scan of the per-thread totals across the block: wave shuffle scan + 4 wave totals in LDS
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int32_t incl = run;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int32_t o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int32_t woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    const int32_t excl = woff + incl - run;
#pragma unroll
    for (int i = 0; i < PXL_SCAN_ITEMS; ++i) {
        int64_t k = base + i;
        if (k < n) rloc[row * n + k] = v[i] + excl;
    }
    if (threadIdx.x == 255) bsum[row * nb + blockIdx.x] = woff + incl;
}

// exclusive scan of the block totals (one block per coordinate row walks them with a running carry)
__global__ __launch_bounds__(1024) void k_scan_bsums(int64_t nb, const int32_t* __restrict__ bsum, int32_t* __restrict__ boff,
                                                     const int32_t* __restrict__ gate) {
    __shared__ int32_t wsum[16];
    __shared__ int32_t carry_s;
    if (gate && *gate == 0) return;
    const int row = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int64_t b0 = 0; b0 < nb; b0 += 1024) {
        int64_t b = b0 + threadIdx.x;
        int32_t x = (b < nb) ? bsum[row * nb + b] : 0;
        int32_t incl = x;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            int32_t o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int32_t woff = 0;
        for (int w = 0; w < wave; ++w) woff += wsum[w];
        const int32_t carry = carry_s;
        if (b < nb) boff[row * nb + b] = carry + woff + incl - x;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + incl;
        __syncthreads();
    }
}

__device__ inline int32_t scan_value(const int32_t* rloc, const int32_t* boff, int64_t n, int64_t nb, int row, int64_t k) {
    return rloc[row * n + k] + boff[row * nb + k / PXL_SCAN_BLOCK];
}

__global__ __launch_bounds__(256) void k_unwrap_verify(int64_t n, int nrow, const double* __restrict__ m2, double period,
                                                       int8_t* __restrict__ c, const int32_t* __restrict__ rloc,
                                                       const int32_t* __restrict__ boff, int64_t nb,
                                                       int32_t* __restrict__ flag, const int32_t* __restrict__ gate) {
    if (gate && *gate == 0) return;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    bool bad = false;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + 1; k < n; k += stride) {
        for (int row = 0; row < nrow; ++row) {
            const int32_t rprev = scan_value(rloc, boff, n, nb, row, k - 1);
            const int32_t rk = scan_value(rloc, boff, n, nb, row, k);
            const double yprev = m2[nrow * (k - 1) + row] - (double)rprev * period;  // y[k-1] as the reference forms it
            const double q = (m2[nrow * k + row] - yprev) / period;
            if (!isfinite(q)) { bad = true; continue; }       // NaN/Inf poison everything after them: serial path
            const int32_t t = (int32_t)rint(q);
            if (t != rk) {
                c[row * n + k] = (int8_t)((int)c[row * n + k] + (t - rk));
                bad = true;
            }
        }
    }
    if (bad) atomicOr(flag, 1);
}

__global__ __launch_bounds__(256) void k_unwrap_apply(int64_t n, int nrow, double* __restrict__ m2, double period, double ref,
                                                      const int32_t* __restrict__ rloc, const int32_t* __restrict__ boff,
                                                      int64_t nb, const int32_t* __restrict__ flag) {
    // flag[0]: pass 1 found mismatches; flag[1]: pass 2 (run only then) still found some
    if (flag[0] && flag[1]) return;         // unverified: the serial kernel produces the answer
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) {
        if (nrow == 2) {
            double2 m = *reinterpret_cast<const double2*>(m2 + 2 * k);
            double y0 = m.x, y1 = m.y;
            if (k > 0) {
                y0 = m.x - (double)scan_value(rloc, boff, n, nb, 0, k) * period;
                y1 = m.y - (double)scan_value(rloc, boff, n, nb, 1, k) * period;
            }
            *reinterpret_cast<double2*>(m2 + 2 * k) = make_double2(y0 + ref, y1 + ref);
        } else {
            double y = m2[k];
            if (k > 0) y = y - (double)scan_value(rloc, boff, n, nb, 0, k) * period;
            m2[k] = y + ref;
        }
    }
}

// Exact serial form (one wave per coordinate row, 64 dependent steps per 64 points): the fallback when the
// speculative scan cannot be verified, and the whole algorithm for tiny batches.  `prewound` = input already
// holds m = rewind(.) - ref.  gate: run only if *gate != 0 (NULL = always).
__global__ __launch_bounds__(64) void k_unwind_rows(int64_t n, int nrow, double* __restrict__ sky, double period, double ref,
                                                    int prewound, const int32_t* __restrict__ gate) {
    if (gate && !(gate[0] && gate[1])) return;
    const int row = blockIdx.x;
    const int lane = threadIdx.x;
    double prev = 0.0;
    bool have_prev = false;
    for (int64_t base = 0; base < n; base += 64) {
        int64_t k = base + lane;
        double m = 0.0;
        if (k < n) m = prewound ? sky[nrow * k + row] : rewind(sky[nrow * k + row], period, ref) - ref;
        double y = m;
        int cnt = (int)((n - base) < 64 ? (n - base) : 64);
        for (int l = 0; l < cnt; ++l) {
            double ml = __shfl(m, l, 64);
            double yl = have_prev ? ml - rint((ml - prev) / period) * period : ml;
            prev = yl;
            have_prev = true;
            if (lane == l) y = yl;
        }
        if (k < n) sky[nrow * k + row] = y + ref;
    }
}

// rewind! on a flat array (enmap_ops.jl:15-19); sub_ref: also subtract ref (first half of unwind!)
__global__ __launch_bounds__(256) void k_rewind(int64_t n, double* a, double period, double ref, int sub_ref) {
    const int64_t chunk = (int64_t)blockDim.x * 4;
    for (int64_t k0 = (int64_t)blockIdx.x * chunk + threadIdx.x; k0 < n; k0 += (int64_t)gridDim.x * chunk) {
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { int64_t k = k0 + u * blockDim.x; v[u] = (k < n) ? a[k] : 0.0; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int64_t k = k0 + u * blockDim.x;
            double r = rewind(v[u], period, ref);
            if (sub_ref) r = r - ref;
            if (k < n) a[k] = r;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// posmap (A15) / pixareamap (N4): write-only maps.  Lane = 2 adjacent RA pixels (16 B stores).
// ------------------------------------------------------------------------------------------------
// Block = (512-column chunk, chunk of rows): RA (rewound) is computed once per lane and reused for every
// row of the chunk; DEC / the row area is one evaluation per row.
#define PXL_POS_ROWS 32
__global__ __launch_bounds__(256) void k_posmap_car(CarAffine c, int64_t nx, int64_t row0, int64_t nrows,
                                                    double* __restrict__ ra, double* __restrict__ dec, int safe) {
    const bool vec = ((nx & 1) == 0) && ((((uintptr_t)ra | (uintptr_t)dec) & 15) == 0);
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;        // 0-based column of the pair
    if (i >= nx) return;
    double a0 = p2s_ra(c, (double)(i + 1));
    double a1 = p2s_ra(c, (double)(i + 2));
    if (safe) { a0 = rewind(a0, PXL_TWOPI_D, 0.0); a1 = rewind(a1, PXL_TWOPI_D, 0.0); }
    const int64_t jr0 = (int64_t)blockIdx.y * PXL_POS_ROWS;
    const int64_t jr1 = (jr0 + PXL_POS_ROWS < nrows) ? jr0 + PXL_POS_ROWS : nrows;
    for (int64_t jr = jr0; jr < jr1; ++jr) {
        double d = p2s_dec(c, (double)(row0 + jr + 1));
        if (safe) d = rewind(d, PXL_TWOPI_D, 0.0);
        int64_t o = jr * nx + i;
        if (vec) {
            *reinterpret_cast<double2*>(ra + o) = make_double2(a0, a1);
            *reinterpret_cast<double2*>(dec + o) = make_double2(d, d);
        } else {
            ra[o] = a0; dec[o] = d;
            if (i + 1 < nx) { ra[o + 1] = a1; dec[o + 1] = d; }
        }
    }
}

__global__ __launch_bounds__(256) void k_pixareamap_car(CarAffine c, int64_t nx, int64_t row0, int64_t nrows,
                                                        double* __restrict__ area) {
    const bool vec = ((nx & 1) == 0) && (((uintptr_t)area & 15) == 0);
    const double da = fabs(c.da);
    // one row per blockIdx.y; the row value is computed once per lane and streamed along RA
    const int64_t jr = blockIdx.y;
    const double row = (double)(row0 + jr + 1);
    // enmap_ops.jl:131-134: dec of the two pixel edges, sorted, clamped to the poles
    double e0 = p2s_dec(c, row - 0.5), e1 = p2s_dec(c, row + 0.5);
    double d1 = fmin(e0, e1), d2 = fmax(e0, e1);
    d1 = fmax(-PXL_PI_D / 2, d1); d2 = fmin(PXL_PI_D / 2, d2);
    const double v = (sin(d2) - sin(d1)) * da;
    const int64_t npair = (nx + 1) / 2;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < npair; t += (int64_t)gridDim.x * blockDim.x) {
        int64_t i = t * 2;
        int64_t o = jr * nx + i;
        if (vec) *reinterpret_cast<double2*>(area + o) = make_double2(v, v);
        else { area[o] = v; if (i + 1 < nx) area[o + 1] = v; }
    }
}

// ------------------------------------------------------------------------------------------------
// Gnomonic (A16), tan_proj.jl:44-75
// ------------------------------------------------------------------------------------------------
struct TanParams { double scale, unit, a0, d0, sd0, cd0, cpx, cpy; };
static TanParams tan_setup(const pxl_car_wcs& w) {
    TanParams t;
    t.scale = 1.0 / w.cdelt[0];
    t.unit = w.unit;
    t.a0 = w.crval[0] * (PXL_PI_D / 180);   // deg2rad.(wcs.crval), tan_proj.jl:47
    t.d0 = w.crval[1] * (PXL_PI_D / 180);
    t.sd0 = sin(t.d0); t.cd0 = cos(t.d0);
    t.cpx = w.crpix[0]; t.cpy = w.crpix[1];
    return t;
}
__device__ inline void tan_sky2pix(const TanParams& t, double a, double d, double* x, double* y) {
    double A = cos(d) * cos(a - t.a0);
    double F = t.scale / t.unit / (t.sd0 * sin(d) + A * t.cd0);
    double LINE = -F * (t.cd0 * sin(d) - A * t.sd0);
    double SAMPLE = -F * cos(d) * sin(a - t.a0);
    *x = t.cpx - SAMPLE;
    *y = t.cpy - LINE;
}
__device__ inline void tan_pix2sky(const TanParams& t, double i, double j, double* a, double* d) {
    double X = (t.cpx - i) * t.unit / t.scale;
    double Y = (t.cpy - j) * t.unit / t.scale;
    double D = atan(sqrt(X * X + Y * Y));
    double B = atan2(-X, Y);
    double sD = sin(D), cD = cos(D), cB = cos(B);
    double XX = t.sd0 * sD * cB + t.cd0 * cD;
    double YY = sD * sin(B);
    *a = t.a0 + atan2(YY, XX);
    *d = asin(t.sd0 * cD - t.cd0 * sD * cB);
}
__global__ __launch_bounds__(256) void k_sky2pix_tan(TanParams t, int64_t n, const double* __restrict__ ra,
                                                     const double* __restrict__ dec, double* __restrict__ x,
                                                     double* __restrict__ y) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride)
        tan_sky2pix(t, ra[k], dec[k], &x[k], &y[k]);
}
__global__ __launch_bounds__(256) void k_pix2sky_tan(TanParams t, int64_t n, const double* __restrict__ ip,
                                                     const double* __restrict__ jp, double* __restrict__ ra,
                                                     double* __restrict__ dec) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride)
        tan_pix2sky(t, ip[k], jp[k], &ra[k], &dec[k]);
}
__global__ __launch_bounds__(256) void k_posmap_tan(TanParams t, int64_t nx, int64_t row0, int64_t nrows,
                                                    double* __restrict__ ra, double* __restrict__ dec) {
    const int64_t total = nx * nrows;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += stride) {
        int64_t jr = k / nx, i = k - jr * nx;
        tan_pix2sky(t, (double)(i + 1), (double)(row0 + jr + 1), &ra[k], &dec[k]);
    }
}

// ------------------------------------------------------------------------------------------------
// Reprojection (R1).
// ------------------------------------------------------------------------------------------------
// Separable tables: for output column i (0-based ic) the source cell xi0[ic] (1-based int) and fraction
// xfx[ic]; same for rows.  (a, d) = pix2sky(out; safe=false) [car_proj.jl:146-147];
// (x, y) = sky2pix(in; safe=true), division form [car_proj.jl:225-231].
__global__ __launch_bounds__(256) void k_build_tables(CarAffine out, Sky2Pix in, int64_t nxo, int64_t nyo,
                                                      int32_t* __restrict__ xi0, double* __restrict__ xfx,
                                                      int32_t* __restrict__ yj0, double* __restrict__ yfy) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nxo + nyo; k += stride) {
        if (k < nxo) {
            double a = p2s_ra(out, (double)(k + 1));
            split_cell(s2p_x(in, a), &xi0[k], &xfx[k]);
        } else {
            int64_t j = k - nxo;
            double d = p2s_dec(out, (double)(j + 1));
            split_cell(s2p_y(in, d), &yj0[j], &yfy[j]);
        }
    }
}

struct ReprojParams {
    const double* src;     // (nx, src_nrows, nc)
    double* dst;           // (nxo, dst_nrows, nc)
    const int32_t* xi0; const double* xfx;   // nxo entries
    const int32_t* yj0; const double* yfy;   // nyo entries (absolute output row)
    int64_t nx, ny, src_row0, src_nrows;
    int64_t nxo, dst_row0, dst_nrows;
    int64_t r0, nr;        // output rows handled by this launch, relative to the dst window
    int32_t nc, periodic;
    // staged kernel only
    int32_t rh;            // output rows per tile
    int32_t seg;           // LDS slot length in doubles (even)
    int32_t dxpos;         // source column increases with output column
    int32_t dypos;         // source row increases with output row
    int32_t ntx, nty;      // tiles along RA / DEC
    int64_t ntiles, tiles_per_xcd;
    int32_t flags;         // tuning/diagnostics: 1 = skip source loads, 2 = skip stores, 4 = no XCD remap
    // LDS-DMA kernel only
    int32_t ns, pf;        // ring slots (power of two), prefetch distance in output rows
    const double* zero_page;   // 16 bytes of zeros in device memory
};

// ---- generic direct-gather kernel: one lane per output pixel pair, 4 taps from global memory each.
//      Used when a tile's source footprint does not fit the LDS ring (large down-scaling) and as the
//      cross-check variant.
__global__ __launch_bounds__(256) void k_reproject_gather(ReprojParams p) {
    const int64_t npair = (p.nxo + 1) / 2;
    const int64_t total = npair * p.nr;
    const int c = blockIdx.y;
    SrcView m{p.src + (int64_t)c * p.nx * p.src_nrows, p.nx, p.ny, p.src_row0, p.src_nrows, p.periodic};
    double* dplane = p.dst + (int64_t)c * p.nxo * p.dst_nrows;
    const bool vec = ((p.nxo & 1) == 0) && (((uintptr_t)p.dst & 15) == 0);
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        int64_t rr = t / npair;
        int64_t i = (t - rr * npair) * 2;
        int64_t r = p.r0 + rr;
        int64_t j0 = p.yj0[p.dst_row0 + r];
        double fy = p.yfy[p.dst_row0 + r];
        double v0 = bilerp_cells(m, p.xi0[i], p.xfx[i], j0, fy);
        int64_t o = r * p.nxo + i;
        if (i + 1 < p.nxo) {
            double v1 = bilerp_cells(m, p.xi0[i + 1], p.xfx[i + 1], j0, fy);
            if (vec) *reinterpret_cast<double2*>(dplane + o) = make_double2(v0, v1);
            else { dplane[o] = v0; dplane[o + 1] = v1; }
        } else {
            dplane[o] = v0;
        }
    }
}

// ---- staged kernel: ONE WAVEFRONT PER OUTPUT TILE.
//
// A tile is TW = 128*PAIRS output columns x rh output rows of one component plane.  The wave marches
// down the tile's rows.  The two source rows an output row needs (j0, j0+1) live in a 4-slot LDS ring
// (slot = row & 3, tagged with the row id), each slot holding the contiguous source-column segment the
// tile's columns touch, loaded with 16 B/lane coalesced reads; the RA seam of a full-sky map is
// resolved while staging (segment column u -> u mod nx), so the interpolation itself never sees it.
// Rows for output row r+1 are prefetched into registers while row r is computed and stored.
// Output is written with 16 B/lane coalesced stores (lane = 2 adjacent RA pixels per PAIR).
//
// Tiles whose columns do not fit the slot (the rewind discontinuity of a partial-sky source falling
// inside the tile) fall back, wave-uniformly, to direct taps.
#define PXL_NS 4
#define PXL_MAXCH 5     // 16-B chunks of 64 lanes per slot: slot <= 5*128 doubles

template <bool VEC>
__device__ inline void load_row_regs(const ReprojParams& p, const double* plane, int64_t j, int64_t cbase0,
                                     int lane, double2 (&regs)[PXL_MAXCH]) {
    // j: 1-based absolute source row (any integer).  Rows outside the map / resident window read as 0.
    int64_t jr = j - 1 - p.src_row0;
    const bool row_ok = (j >= 1) && (j <= p.ny) && (jr >= 0) && (jr < p.src_nrows) && !(p.flags & 1);
    const double* rowp = plane + (row_ok ? jr : 0) * p.nx;
#pragma unroll
    for (int ch = 0; ch < PXL_MAXCH; ++ch) {
        int k = ch * 128 + 2 * lane;
        double2 v = make_double2(0.0, 0.0);
        if (k < p.seg && row_ok) {
            int64_t u = cbase0 + k;                   // 0-based unwrapped column of the chunk's first element
            if (VEC) {
                // nx even and u even: the pair never straddles the seam or the map edge
                bool ok = true;
                if (p.periodic) { u %= p.nx; if (u < 0) u += p.nx; }
                else ok = (u >= 0) && (u < p.nx);
                if (ok) v = *reinterpret_cast<const double2*>(rowp + u);
            } else {
                int64_t u0 = u, u1 = u + 1;
                bool ok0 = true, ok1 = true;
                if (p.periodic) {
                    u0 %= p.nx; if (u0 < 0) u0 += p.nx;
                    u1 %= p.nx; if (u1 < 0) u1 += p.nx;
                } else {
                    ok0 = (u0 >= 0) && (u0 < p.nx);
                    ok1 = (u1 >= 0) && (u1 < p.nx);
                }
                if (ok0) v.x = rowp[u0];
                if (ok1) v.y = rowp[u1];
            }
        }
        regs[ch] = v;
    }
}

__device__ inline void store_row_lds(const ReprojParams& p, double* slot, int lane, const double2 (&regs)[PXL_MAXCH]) {
#pragma unroll
    for (int ch = 0; ch < PXL_MAXCH; ++ch) {
        int k = ch * 128 + 2 * lane;
        if (k < p.seg) *reinterpret_cast<double2*>(slot + k) = regs[ch];
    }
}

template <int PAIRS, bool VEC>
__global__ __launch_bounds__(64) void k_reproject_staged(ReprojParams p) {
    extern __shared__ __attribute__((aligned(16))) double lds[];   // PXL_NS * seg doubles
    const int lane = threadIdx.x;
    constexpr int TW = 128 * PAIRS;

    // XCD-aware decode: hardware deals blocks round-robin over the 8 XCDs (b % 8); give each XCD a
    // contiguous run of tiles so RA-neighbouring tiles (which share 128-B lines at their edges and the
    // same source rows) hit the same L2.  Placement only affects speed, never correctness.
    const int64_t b = blockIdx.x;
    const int64_t t = (p.flags & 4) ? b : (b & 7) * p.tiles_per_xcd + (b >> 3);
    if (t >= p.ntiles) return;
    const int tx = (int)(t % p.ntx);
    const int64_t trest = t / p.ntx;
    const int ty = (int)(trest % p.nty);
    const int c = (int)(trest / p.nty);

    const double* splane = p.src + (int64_t)c * p.nx * p.src_nrows;
    double* dplane = p.dst + (int64_t)c * p.nxo * p.dst_nrows;

    const int64_t c0 = (int64_t)tx * TW;                       // first output column of the tile
    const int64_t clast = (c0 + TW < p.nxo ? c0 + TW : p.nxo) - 1;
    const int64_t rb = p.r0 + (int64_t)ty * p.rh;              // rows relative to the dst window
    const int64_t re = (rb + p.rh < p.r0 + p.nr) ? rb + p.rh : p.r0 + p.nr;

    // ---- per-lane column setup
    const int64_t a = p.dxpos ? p.xi0[c0] : p.xi0[clast];      // 1-based source cell of the tile's low end
    const int64_t ua = a - 1;
    const int64_t cbase0 = ua & ~(int64_t)1;                   // even 0-based column at slot index 0
    int dloc[PAIRS][2];
    double fx[PAIRS][2];
    bool act[PAIRS][2];
    bool fits = true;
#pragma unroll
    for (int q = 0; q < PAIRS; ++q) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            int64_t col = c0 + q * 128 + 2 * lane + e;
            act[q][e] = col < p.nxo;
            int64_t i0 = act[q][e] ? p.xi0[col] : a;
            fx[q][e] = act[q][e] ? p.xfx[col] : 0.0;
            int64_t d = i0 - a;
            if (p.periodic) { d %= p.nx; if (d < 0) d += p.nx; }
            d += ua - cbase0;
            if (d < 0 || d + 1 >= p.seg) fits = false;
            dloc[q][e] = (int)d;
        }
    }
    const bool vec_store = ((p.nxo & 1) == 0) && (((uintptr_t)p.dst & 15) == 0);

    if (!__all(fits)) {
        // wave-uniform fallback: direct taps for this tile
        SrcView m{splane, p.nx, p.ny, p.src_row0, p.src_nrows, p.periodic};
        for (int64_t r = rb; r < re; ++r) {
            int64_t j0 = p.yj0[p.dst_row0 + r];
            double fy = p.yfy[p.dst_row0 + r];
#pragma unroll
            for (int q = 0; q < PAIRS; ++q)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    int64_t col = c0 + q * 128 + 2 * lane + e;
                    if (act[q][e]) dplane[r * p.nxo + col] = bilerp_cells(m, p.xi0[col], fx[q][e], j0, fy);
                }
        }
        return;
    }

    // ---- ring state (wave-uniform): tag of the source row held by each slot
    int64_t tag0 = INT64_MIN, tag1 = INT64_MIN, tag2 = INT64_MIN, tag3 = INT64_MIN;
    auto resident = [&](int64_t j) -> bool {
        int s = (int)(j & 3);
        int64_t tg = (s == 0) ? tag0 : (s == 1) ? tag1 : (s == 2) ? tag2 : tag3;
        return tg == j;
    };
    auto settag = [&](int64_t j) {
        int s = (int)(j & 3);
        if (s == 0) tag0 = j; else if (s == 1) tag1 = j; else if (s == 2) tag2 = j; else tag3 = j;
    };

    // per-row table entries of this tile live in lane (r - rb) and are broadcast with v_readlane
    // (no scalar-memory round trip inside the row loop); rh <= 64
    int my_j0 = 0;
    double my_fy = 0.0;
    if (rb + lane < re) { my_j0 = p.yj0[p.dst_row0 + rb + lane]; my_fy = p.yfy[p.dst_row0 + rb + lane]; }
    auto row_j0 = [&](int64_t r) -> int64_t { return (int64_t)__builtin_amdgcn_readlane(my_j0, (int)(r - rb)); };
    auto row_fy = [&](int64_t r) -> double {
        int lo = __builtin_amdgcn_readlane(__double2loint(my_fy), (int)(r - rb));
        int hi = __builtin_amdgcn_readlane(__double2hiint(my_fy), (int)(r - rb));
        return __hiloint2double(hi, lo);
    };

    double2 ra_[PXL_MAXCH], rb_[PXL_MAXCH];
    {   // prologue: rows of the first output row
        int64_t j0 = row_j0(rb);
        load_row_regs<VEC>(p, splane, j0, cbase0, lane, ra_);
        load_row_regs<VEC>(p, splane, j0 + 1, cbase0, lane, rb_);
        store_row_lds(p, lds + (j0 & 3) * p.seg, lane, ra_);
        store_row_lds(p, lds + ((j0 + 1) & 3) * p.seg, lane, rb_);
        settag(j0); settag(j0 + 1);
        __syncthreads();
    }

    for (int64_t r = rb; r < re; ++r) {
        const int64_t j0 = row_j0(r);
        const double fy = row_fy(r);

        // prefetch the rows output row r+1 needs and the ring lacks (global -> registers)
        bool needA = false, needB = false;
        int64_t jn = 0;
        if (r + 1 < re) {
            jn = row_j0(r + 1);
            needA = !resident(jn);
            needB = !resident(jn + 1);
            if (needA) load_row_regs<VEC>(p, splane, jn, cbase0, lane, ra_);
            if (needB) load_row_regs<VEC>(p, splane, jn + 1, cbase0, lane, rb_);
        }

        // interpolate output row r from LDS
        const double* T = lds + (j0 & 3) * p.seg;
        const double* B = lds + ((j0 + 1) & 3) * p.seg;
        const double wy = 1 - fy;
#pragma unroll
        for (int q = 0; q < PAIRS; ++q) {
            double v[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                int d = dloc[q][e];
                double wx = 1 - fx[q][e];
                double top = wx * T[d] + fx[q][e] * T[d + 1];
                double bot = wx * B[d] + fx[q][e] * B[d + 1];
                v[e] = wy * top + fy * bot;
            }
            int64_t col = c0 + q * 128 + 2 * lane;
            double* o = dplane + r * p.nxo + col;
            if (p.flags & 2) { if (v[0] == 1.2345e300) o[0] = v[1]; }       // diagnostics: keep v live, never store
            else if (vec_store) { if (act[q][0]) *reinterpret_cast<double2*>(o) = make_double2(v[0], v[1]); }
            else { if (act[q][0]) o[0] = v[0]; if (act[q][1]) o[1] = v[1]; }
        }

        if (needA || needB) {
            __syncthreads();                       // every lane is done reading the slots being replaced
            if (needA) { store_row_lds(p, lds + (jn & 3) * p.seg, lane, ra_); settag(jn); }
            if (needB) { store_row_lds(p, lds + ((jn + 1) & 3) * p.seg, lane, rb_); settag(jn + 1); }
            __syncthreads();
        }
    }
}


#include "pxl_reproject_dma.h"

// ---- generic (non-separable) bilinear reprojection between CAR and Gnomonic maps (N2).
// Per output pixel: (ra, dec) = pix2sky(out) [car_proj.jl:146-147 safe=false | tan_proj.jl:59-75];
// (x, y) = sky2pix(in) [car_proj.jl:225-231 safe=true | tan_proj.jl:44-57]; 2x2 direct taps + lerp.
// A sky point behind a Gnomonic source's tangent plane (cos c <= 0) is not on that map: it reads as 0.
// FP64-transcendental bound (about ten libm calls per pixel), tolerance-checked rather than bit-exact.
struct GenericParams {
    const double* src; double* dst;
    int64_t nx, ny, nxo, nyo;
    int32_t nc, periodic, proj_in, proj_out;
    CarAffine out_car; TanParams out_tan;
    Sky2Pix in_car; TanParams in_tan;
};
__global__ __launch_bounds__(256) void k_reproject_generic(GenericParams p) {
    const int64_t total = p.nxo * p.nyo;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const int64_t jr = t / p.nxo, i = t - jr * p.nxo;
        double ra, dec;
        if (p.proj_out == PXL_PROJ_TAN) tan_pix2sky(p.out_tan, (double)(i + 1), (double)(jr + 1), &ra, &dec);
        else { ra = p2s_ra(p.out_car, (double)(i + 1)); dec = p2s_dec(p.out_car, (double)(jr + 1)); }
        double x, y;
        bool visible = true;
        if (p.proj_in == PXL_PROJ_TAN) {
            tan_sky2pix(p.in_tan, ra, dec, &x, &y);
            visible = (p.in_tan.sd0 * sin(dec) + cos(dec) * cos(ra - p.in_tan.a0) * p.in_tan.cd0) > 0.0;
        } else { x = s2p_x(p.in_car, ra); y = s2p_y(p.in_car, dec); }
        const bool fin = isfinite(x) && isfinite(y);
        int32_t i0, j0; double fx, fy;
        split_cell(x, &i0, &fx);
        split_cell(y, &j0, &fy);
        for (int c = 0; c < p.nc; ++c) {
            SrcView m{p.src + (int64_t)c * p.nx * p.ny, p.nx, p.ny, 0, p.ny, p.periodic};
            double v = visible ? bilerp_cells(m, i0, fx, j0, fy) : 0.0;
            p.dst[(int64_t)c * total + t] = fin ? v : __builtin_nan("");
        }
    }
}

// ---- scattered sample: fused sky2pix!(safe=true) [car_proj.jl:165-193] + 2x2 gather + lerp.
// An irregular gather: each point touches two 16-byte spans in two different rows of a multi-GB map, so
// the kernel is bound by random-sector fetches, not by bytes.  Each lane handles PXL_SUNR points per trip
// and issues all their taps before any arithmetic (4x the gathers in flight per lane); tap indices are
// 32-bit and the RA wrap is one conditional add/subtract (safe sky2pix keeps x within half a period of the
// map centre), with the oracle's full modulo kept only as the out-of-range path.
#define PXL_SUNR 4
__device__ inline int64_t wrap_col(int64_t i, int64_t nx) {          // 1-based column of a periodic map
    if (i >= 1 - nx && i <= 2 * nx) { if (i > nx) i -= nx; else if (i < 1) i += nx; return i; }
    i = (i - 1) % nx; if (i < 0) i += nx; return i + 1;
}
__global__ __launch_bounds__(256) void k_sample_bilinear(Sky2Pix s, const double* __restrict__ src, int64_t nx,
                                                         int64_t ny, int32_t nc, int64_t row0, int64_t nrows,
                                                         int periodic, int64_t n, const double2* __restrict__ sky,
                                                         double* __restrict__ out) {
    const int64_t chunk = (int64_t)blockDim.x * PXL_SUNR;
    const int64_t plane = nx * nrows;
    for (int64_t k0 = (int64_t)blockIdx.x * chunk + threadIdx.x; k0 < n; k0 += (int64_t)gridDim.x * chunk) {
        double2 ad[PXL_SUNR];
#pragma unroll
        for (int u = 0; u < PXL_SUNR; ++u) {
            int64_t k = k0 + u * blockDim.x;
            ad[u] = (k < n) ? sky[k] : make_double2(0.0, 0.0);
        }
        int64_t o00[PXL_SUNR], o10[PXL_SUNR], o01[PXL_SUNR], o11[PXL_SUNR];   // element offsets, -1 = reads as 0
        double fx[PXL_SUNR], fy[PXL_SUNR];
        bool fin[PXL_SUNR];
#pragma unroll
        for (int u = 0; u < PXL_SUNR; ++u) {
            double x = s2p_x(s, ad[u].x), y = s2p_y(s, ad[u].y);
            fin[u] = isfinite(x) && isfinite(y);
            int32_t i0, j0;
            split_cell(x, &i0, &fx[u]);
            split_cell(y, &j0, &fy[u]);
            int64_t ia = i0, ib = (int64_t)i0 + 1;
            bool oka = true, okb = true;
            if (periodic) { ia = wrap_col(ia, nx); ib = wrap_col(ib, nx); }
            else { oka = (ia >= 1 && ia <= nx); okb = (ib >= 1 && ib <= nx); }
            int64_t ja = (int64_t)j0 - 1 - row0, jb = ja + 1;                    // resident row indices
            bool rowa = (j0 >= 1 && j0 <= ny && ja >= 0 && ja < nrows);
            bool rowb = ((int64_t)j0 + 1 >= 1 && (int64_t)j0 + 1 <= ny && jb >= 0 && jb < nrows);
            o00[u] = (rowa && oka) ? ja * nx + (ia - 1) : -1;
            o10[u] = (rowa && okb) ? ja * nx + (ib - 1) : -1;
            o01[u] = (rowb && oka) ? jb * nx + (ia - 1) : -1;
            o11[u] = (rowb && okb) ? jb * nx + (ib - 1) : -1;
        }
        for (int c = 0; c < nc; ++c) {
            const double* pl = src + (int64_t)c * plane;
            double m00[PXL_SUNR], m10[PXL_SUNR], m01[PXL_SUNR], m11[PXL_SUNR];
#pragma unroll
            for (int u = 0; u < PXL_SUNR; ++u) {
                m00[u] = o00[u] >= 0 ? pl[o00[u]] : 0.0;
                m10[u] = o10[u] >= 0 ? pl[o10[u]] : 0.0;
                m01[u] = o01[u] >= 0 ? pl[o01[u]] : 0.0;
                m11[u] = o11[u] >= 0 ? pl[o11[u]] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < PXL_SUNR; ++u) {
                int64_t k = k0 + u * blockDim.x;
                double top = (1 - fx[u]) * m00[u] + fx[u] * m10[u];
                double bot = (1 - fx[u]) * m01[u] + fx[u] * m11[u];
                double v = (1 - fy[u]) * top + fy[u] * bot;
                if (k < n) out[(int64_t)c * n + k] = fin[u] ? v : __builtin_nan("");
            }
        }
    }
}

// ---- FITS staging (N3): big-endian image data <-> native Float64, on the device (enmap.jl:198-237 reads and
// writes BITPIX -64/-32 image HDUs through CFITSIO; here the raw bytes are copied to HBM and swapped there).
__global__ __launch_bounds__(256) void k_bswap_to_f64(const void* raw, double* dst,
                                                      int64_t n, int bitpix) {
    const int64_t chunk = (int64_t)blockDim.x * 4;
    for (int64_t k0 = (int64_t)blockIdx.x * chunk + threadIdx.x; k0 < n; k0 += (int64_t)gridDim.x * chunk) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int64_t k = k0 + u * blockDim.x;
            if (k >= n) continue;
            if (bitpix == -64) {
                uint64_t v = __builtin_bswap64(reinterpret_cast<const uint64_t*>(raw)[k]);
                dst[k] = __longlong_as_double((long long)v);
            } else {            // -32: IEEE single, widened exactly
                uint32_t v = __builtin_bswap32(reinterpret_cast<const uint32_t*>(raw)[k]);
                dst[k] = (double)__uint_as_float(v);
            }
        }
    }
}
__global__ __launch_bounds__(256) void k_f64_to_be(const double* __restrict__ src, uint64_t* __restrict__ raw, int64_t n) {
    const int64_t chunk = (int64_t)blockDim.x * 4;
    for (int64_t k0 = (int64_t)blockIdx.x * chunk + threadIdx.x; k0 < n; k0 += (int64_t)gridDim.x * chunk) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int64_t k = k0 + u * blockDim.x;
            if (k < n) raw[k] = __builtin_bswap64((uint64_t)__double_as_longlong(src[k]));
        }
    }
}

// ---- synthetic data (benchmark plumbing): splitmix64 counter RNG
__device__ inline uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ inline double u01(uint64_t bits) { return (double)(bits >> 11) * (1.0 / 9007199254740992.0); }

__global__ __launch_bounds__(256) void k_fill_random(double* __restrict__ dst, int64_t n, uint64_t seed,
                                                     uint64_t offset, int kind) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) {
        uint64_t ctr = (uint64_t)k + offset;
        uint64_t h1 = splitmix64(seed ^ splitmix64(2 * ctr));
        double u1 = u01(h1);
        if (kind == 1) { dst[k] = u1; continue; }
        uint64_t h2 = splitmix64(seed ^ splitmix64(2 * ctr + 1));
        double u2 = u01(h2);
        dst[k] = sqrt(-2.0 * log(1.0 - u1)) * cos(PXL_TWOPI_D * u2);   // Box-Muller
    }
}
__global__ __launch_bounds__(256) void k_fill_sphere(double2* __restrict__ sky, int64_t n, uint64_t seed,
                                                     uint64_t offset) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) {
        uint64_t ctr = (uint64_t)k + offset;
        double u1 = u01(splitmix64(seed ^ splitmix64(2 * ctr)));
        double u2 = u01(splitmix64(seed ^ splitmix64(2 * ctr + 1)));
        sky[k] = make_double2(PXL_TWOPI_D * u1 - PXL_PI_D, asin(2.0 * u2 - 1.0));
    }
}

// ================================================================================================
// C ABI
// ================================================================================================
struct pxl_reproject_plan {
    pxl_car_wcs win, wout;
    int64_t nx, ny, nc, src_row0, src_nrows;
    int64_t nxo, nyo, dst_row0, dst_nrows;
    int periodic;
    int device;
    // device tables
    int32_t* xi0; double* xfx; int32_t* yj0; double* yfy;
    void* table_mem;
    // host copy of the row table (cells only), same arithmetic as the device
    int32_t* h_yj0;
    // launch configuration
    int variant;       // 0 auto, 1 gather, 2 staged
    int pairs;         // lane width of the register-staged kernel: 1 or 2 (x2 output columns per lane)
    int pairs_dma;     // lane width of the LDS-DMA kernel: 1, 2 or 4
    int seg_dma;
    int dypos;
    int rh;
    int seg;
    int dxpos;
    int flags;
    int ns, pf;
    double* zero_page;
    bool staged_ok;
    bool vec_load;
    bool tables_built;
};

// unwind!(sky2xN; dims=2) on a buffer that already holds m = rewind(.) - ref (see k_unwrap_* above).
// Scratch comes from the stream-ordered allocator (hipMallocAsync / hipFreeAsync): no host synchronisation.
static int unwind_rows(int64_t n, int nrow, double* sky, double period, double ref, hipStream_t st) {
    if (n <= 4096) {       // tiny batches: the exact serial kernel is already fast enough
        hipLaunchKernelGGL(k_unwind_rows, dim3(nrow), dim3(64), 0, st, n, nrow, sky, period, ref, 1, (const int32_t*)nullptr);
        return check_launch("k_unwind_rows");
    }
    const int64_t nb = (n + PXL_SCAN_BLOCK - 1) / PXL_SCAN_BLOCK;
    if (nb > 0x7fffffffLL) return fail(PXL_EINVAL, "unwind: batch too long");
    const size_t bytes_c = (size_t)nrow * n, bytes_r = (size_t)4 * nrow * n, bytes_b = (size_t)4 * nrow * nb;
    const size_t off_r = (bytes_c + 255) & ~(size_t)255;
    const size_t off_bs = off_r + ((bytes_r + 255) & ~(size_t)255);
    const size_t off_bo = off_bs + ((bytes_b + 255) & ~(size_t)255);
    const size_t off_fl = off_bo + ((bytes_b + 255) & ~(size_t)255);
    char* ws = nullptr;
    HIP_TRY(hipMallocAsync((void**)&ws, off_fl + 256, st));
    int8_t* c = (int8_t*)ws;
    int32_t* rloc = (int32_t*)(ws + off_r);
    int32_t* bsum = (int32_t*)(ws + off_bs);
    int32_t* boff = (int32_t*)(ws + off_bo);
    int32_t* flag = (int32_t*)(ws + off_fl);
    int rc = PXL_OK;
    const unsigned g = stream_grid(n, 256);
    hipLaunchKernelGGL(k_unwrap_incr, dim3(g), dim3(256), 0, st, n, nrow, (const double*)sky, period, c);
    if (hipMemsetAsync(flag, 0, 2 * sizeof(int32_t), st) != hipSuccess) rc = fail(PXL_EHIP, "unwind: hipMemsetAsync failed");
    for (int pass = 0; pass < 2 && rc == PXL_OK; ++pass) {
        const int32_t* gate = pass == 0 ? nullptr : flag;        // pass 2 runs on the device only if pass 1 flagged
        hipLaunchKernelGGL((k_scan_local<int8_t>), dim3((unsigned)nb, nrow), dim3(256), 0, st, n, (const int8_t*)c, rloc, bsum, nb, gate);
        hipLaunchKernelGGL(k_scan_bsums, dim3(nrow), dim3(1024), 0, st, nb, (const int32_t*)bsum, boff, gate);
        hipLaunchKernelGGL(k_unwrap_verify, dim3(g), dim3(256), 0, st, n, nrow, (const double*)sky, period, c,
                           (const int32_t*)rloc, (const int32_t*)boff, nb, flag + pass, gate);
        rc = check_launch("k_unwrap scan/verify");
    }
    if (rc == PXL_OK) {
        hipLaunchKernelGGL(k_unwrap_apply, dim3(g), dim3(256), 0, st, n, nrow, sky, period, ref, (const int32_t*)rloc,
                           (const int32_t*)boff, nb, (const int32_t*)flag);
        hipLaunchKernelGGL(k_unwind_rows, dim3(nrow), dim3(64), 0, st, n, nrow, sky, period, ref, 1, (const int32_t*)flag);
        rc = check_launch("k_unwrap_apply");
    }
    hipError_t e = hipFreeAsync(ws, st);
    if (e != hipSuccess && rc == PXL_OK) rc = fail(PXL_EHIP, "unwind: hipFreeAsync: %s", hipGetErrorString(e));
    return rc;
}

extern "C" {

int pxl_version(void) { return PXL_VERSION; }

size_t pxl_last_error(char* buf, size_t n) {
    size_t len = strlen(g_err);
    if (buf && n) {
        size_t m = len < n - 1 ? len : n - 1;
        memcpy(buf, g_err, m);
        buf[m] = 0;
    }
    return len;
}

int pxl_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(PXL_ENODEV, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

int pxl_pix2sky_car_f64(const pxl_car_wcs* wcs, int64_t n, const double* pix, double* sky, int wrap_mode,
                        void* stream) {
    if (!wcs_ok(wcs)) return fail(PXL_EINVAL, "pix2sky: invalid WCS");
    if (n < 0 || (n > 0 && (!pix || !sky))) return fail(PXL_EINVAL, "pix2sky: null buffer or negative n");
    if (wrap_mode < PXL_WRAP_NONE || wrap_mode > PXL_WRAP_UNWIND) return fail(PXL_EINVAL, "pix2sky: bad wrap_mode %d", wrap_mode);
    if ((((uintptr_t)pix | (uintptr_t)sky) & 15) != 0) return fail(PXL_EINVAL, "pix2sky: 2xN buffers must be 16-byte aligned");
    if (n == 0) return PXL_OK;
    hipStream_t st = (hipStream_t)stream;
    CarAffine c = car_affine(*wcs);
    const int mode = wrap_mode == PXL_WRAP_REWIND ? 1 : (wrap_mode == PXL_WRAP_UNWIND ? 2 : 0);
    hipLaunchKernelGGL(k_pix2sky_pairs, dim3(stream_grid((n + PXL_UNR - 1) / PXL_UNR, 256)), dim3(256), 0, st, c, n,
                       (const double2*)pix, (double2*)sky, mode);
    int rc = check_launch("k_pix2sky_pairs");
    if (rc || wrap_mode != PXL_WRAP_UNWIND) return rc;
    return unwind_rows(n, 2, sky, PXL_TWOPI_D, 0.0, st);
}

int pxl_rewind_f64(double* a, int64_t n, double period, double ref_angle, void* stream) {
    if (n < 0 || (n > 0 && !a)) return fail(PXL_EINVAL, "rewind: null buffer or negative n");
    if (!(period > 0.0) || !std::isfinite(period) || !std::isfinite(ref_angle)) return fail(PXL_EINVAL, "rewind: period must be positive and finite");
    if (n == 0) return PXL_OK;
    hipLaunchKernelGGL(k_rewind, dim3(stream_grid((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, n, a, period, ref_angle, 0);
    return check_launch("k_rewind");
}

int pxl_unwind_f64(double* a, int64_t n, int nrow, double period, double ref_angle, void* stream) {
    if (n < 0 || (n > 0 && !a)) return fail(PXL_EINVAL, "unwind: null buffer or negative n");
    if (nrow != 1 && nrow != 2) return fail(PXL_EINVAL, "unwind: nrow must be 1 (vector) or 2 (2xN batch)");
    if (!(period > 0.0) || !std::isfinite(period) || !std::isfinite(ref_angle)) return fail(PXL_EINVAL, "unwind: period must be positive and finite");
    if (nrow == 2 && ((uintptr_t)a & 15) != 0) return fail(PXL_EINVAL, "unwind: 2xN buffer must be 16-byte aligned");
    if (n == 0) return PXL_OK;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_rewind, dim3(stream_grid(((int64_t)nrow * n + 3) / 4, 256)), dim3(256), 0, st, (int64_t)nrow * n, a, period, ref_angle, 1);
    int rc = check_launch("k_rewind");
    if (rc) return rc;
    return unwind_rows(n, nrow, a, period, ref_angle, st);
}

int pxl_pix2sky_car_soa_f64(const pxl_car_wcs* wcs, int64_t n, const double* ipix, const double* jpix,
                            double* ra, double* dec, int safe, void* stream) {
    if (!wcs_ok(wcs)) return fail(PXL_EINVAL, "pix2sky_soa: invalid WCS");
    if (n < 0 || (n > 0 && (!ipix || !jpix || !ra || !dec))) return fail(PXL_EINVAL, "pix2sky_soa: null buffer or negative n");
    if (n == 0) return PXL_OK;
    hipLaunchKernelGGL(k_pix2sky_soa, dim3(stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       car_affine(*wcs), n, ipix, jpix, ra, dec, safe ? 1 : 0);
    return check_launch("k_pix2sky_soa");
}

int pxl_sky2pix_car_f64(const pxl_car_wcs* wcs, const int64_t shape[2], int64_t n, const double* sky,
                        double* pix, int safe, int form, void* stream) {
    if (!wcs_ok(wcs) || !shape) return fail(PXL_EINVAL, "sky2pix: invalid WCS/shape");
    if (n < 0 || (n > 0 && (!pix || !sky))) return fail(PXL_EINVAL, "sky2pix: null buffer or negative n");
    if (form < 0 || form > 2) return fail(PXL_EINVAL, "sky2pix: bad form %d", form);
    if ((((uintptr_t)pix | (uintptr_t)sky) & 15) != 0) return fail(PXL_EINVAL, "sky2pix: 2xN buffers must be 16-byte aligned");
    if (n == 0) return PXL_OK;
    Sky2Pix s = sky2pix_setup(*wcs, shape[0], shape[1], safe ? 1 : 0, form);
    hipLaunchKernelGGL(k_sky2pix_pairs, dim3(stream_grid((n + PXL_UNR - 1) / PXL_UNR, 256)), dim3(256), 0,
                       (hipStream_t)stream, s, n, (const double2*)sky, (double2*)pix);
    return check_launch("k_sky2pix_pairs");
}

int pxl_sky2pix_car_soa_f64(const pxl_car_wcs* wcs, const int64_t shape[2], int64_t n, const double* ra,
                            const double* dec, double* ipix, double* jpix, int safe, int form, void* stream) {
    if (!wcs_ok(wcs) || !shape) return fail(PXL_EINVAL, "sky2pix_soa: invalid WCS/shape");
    if (n < 0 || (n > 0 && (!ipix || !jpix || !ra || !dec))) return fail(PXL_EINVAL, "sky2pix_soa: null buffer or negative n");
    if (form < 0 || form > 2) return fail(PXL_EINVAL, "sky2pix_soa: bad form %d", form);
    if (n == 0) return PXL_OK;
    Sky2Pix s = sky2pix_setup(*wcs, shape[0], shape[1], safe ? 1 : 0, form);
    hipLaunchKernelGGL(k_sky2pix_soa, dim3(stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, s, n, ra,
                       dec, ipix, jpix);
    return check_launch("k_sky2pix_soa");
}

static int check_rows(const char* who, const int64_t shape[2], int64_t row0, int64_t nrows) {
    if (!shape || shape[0] < 1 || shape[1] < 1) return fail(PXL_EINVAL, "%s: bad shape", who);
    if (row0 < 0 || nrows < 0 || row0 + nrows > shape[1])
        return fail(PXL_EINVAL, "%s: rows [%lld, %lld) outside the map (ny=%lld)", who, (long long)row0,
                    (long long)(row0 + nrows), (long long)shape[1]);
    return PXL_OK;
}

int pxl_posmap_car_f64(const pxl_car_wcs* wcs, const int64_t shape[2], int64_t row0, int64_t nrows,
                       double* ra, double* dec, int safe, void* stream) {
    if (!wcs_ok(wcs)) return fail(PXL_EINVAL, "posmap: invalid WCS");
    int rc = check_rows("posmap", shape, row0, nrows);
    if (rc) return rc;
    if (nrows == 0) return PXL_OK;
    if (!ra || !dec) return fail(PXL_EINVAL, "posmap: null output");
    if (nrows > 65535LL * PXL_POS_ROWS) return fail(PXL_EINVAL, "posmap: more than %lld rows per call", 65535LL * PXL_POS_ROWS);
    dim3 grid((unsigned)(((shape[0] + 1) / 2 + 255) / 256), (unsigned)((nrows + PXL_POS_ROWS - 1) / PXL_POS_ROWS));
    hipLaunchKernelGGL(k_posmap_car, grid, dim3(256), 0, (hipStream_t)stream,
                       car_affine(*wcs), shape[0], row0, nrows, ra, dec, safe ? 1 : 0);
    return check_launch("k_posmap_car");
}

int pxl_pixareamap_car_f64(const pxl_car_wcs* wcs, const int64_t shape[2], int64_t row0, int64_t nrows,
                           double* area, void* stream) {
    if (!wcs_ok(wcs)) return fail(PXL_EINVAL, "pixareamap: invalid WCS");
    int rc = check_rows("pixareamap", shape, row0, nrows);
    if (rc) return rc;
    if (nrows == 0) return PXL_OK;
    if (!area) return fail(PXL_EINVAL, "pixareamap: null output");
    // rows go on grid.y (<= 65535 per launch)
    for (int64_t r = 0; r < nrows; r += 65535) {
        int64_t nr = (nrows - r < 65535) ? nrows - r : 65535;
        unsigned gx = (unsigned)std::max<int64_t>(1, ((shape[0] + 1) / 2 + 2047) / 2048);   // ~8 pairs per lane
        hipLaunchKernelGGL(k_pixareamap_car, dim3(gx, (unsigned)nr), dim3(256), 0, (hipStream_t)stream,
                           car_affine(*wcs), shape[0], row0 + r, nr, area + r * shape[0]);
        int rc2 = check_launch("k_pixareamap_car");
        if (rc2) return rc2;
    }
    return PXL_OK;
}

int pxl_sky2pix_tan_f64(const pxl_car_wcs* wcs, int64_t n, const double* ra, const double* dec, double* ipix,
                        double* jpix, void* stream) {
    if (!wcs_ok(wcs)) return fail(PXL_EINVAL, "sky2pix_tan: invalid WCS");
    if (n < 0 || (n > 0 && (!ipix || !jpix || !ra || !dec))) return fail(PXL_EINVAL, "sky2pix_tan: null buffer or negative n");
    if (n == 0) return PXL_OK;
    hipLaunchKernelGGL(k_sky2pix_tan, dim3(stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       tan_setup(*wcs), n, ra, dec, ipix, jpix);
    return check_launch("k_sky2pix_tan");
}

int pxl_pix2sky_tan_f64(const pxl_car_wcs* wcs, int64_t n, const double* ipix, const double* jpix, double* ra,
                        double* dec, void* stream) {
    if (!wcs_ok(wcs)) return fail(PXL_EINVAL, "pix2sky_tan: invalid WCS");
    if (n < 0 || (n > 0 && (!ipix || !jpix || !ra || !dec))) return fail(PXL_EINVAL, "pix2sky_tan: null buffer or negative n");
    if (n == 0) return PXL_OK;
    hipLaunchKernelGGL(k_pix2sky_tan, dim3(stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       tan_setup(*wcs), n, ipix, jpix, ra, dec);
    return check_launch("k_pix2sky_tan");
}

int pxl_posmap_tan_f64(const pxl_car_wcs* wcs, const int64_t shape[2], int64_t row0, int64_t nrows, double* ra,
                       double* dec, void* stream) {
    if (!wcs_ok(wcs)) return fail(PXL_EINVAL, "posmap_tan: invalid WCS");
    int rc = check_rows("posmap_tan", shape, row0, nrows);
    if (rc) return rc;
    if (nrows == 0) return PXL_OK;
    if (!ra || !dec) return fail(PXL_EINVAL, "posmap_tan: null output");
    hipLaunchKernelGGL(k_posmap_tan, dim3(stream_grid(shape[0] * nrows, 256)), dim3(256), 0,
                       (hipStream_t)stream, tan_setup(*wcs), shape[0], row0, nrows, ra, dec);
    return check_launch("k_posmap_tan");
}

// ---- reprojection plan -------------------------------------------------------------------------
static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

int pxl_reproject_plan_create(const pxl_car_wcs* wcs_in, const int64_t shape_in[3], int64_t src_row0,
                              int64_t src_nrows, const pxl_car_wcs* wcs_out, const int64_t shape_out[2],
                              int64_t dst_row0, int64_t dst_nrows, pxl_reproject_plan** out) {
    if (!out) return fail(PXL_EINVAL, "plan_create: null plan pointer");
    *out = nullptr;
    if (!wcs_ok(wcs_in) || !wcs_ok(wcs_out)) return fail(PXL_EINVAL, "plan_create: invalid WCS");
    if (!shape_in || !shape_out) return fail(PXL_EINVAL, "plan_create: null shape");
    if (shape_in[0] < 1 || shape_in[1] < 1 || shape_in[2] < 1 || shape_out[0] < 1 || shape_out[1] < 1)
        return fail(PXL_EINVAL, "plan_create: shapes must be positive");
    // int32 cell tables and 32-bit byte offsets within a source row: nx * 8 must stay below 2^32
    if (shape_in[0] > 400000000 || shape_in[1] > 1000000000 || shape_out[0] > 1000000000 || shape_out[1] > 1000000000)
        return fail(PXL_EINVAL, "plan_create: axis too long (source RA axis <= 4e8, others <= 1e9 pixels)");
    if (src_row0 < 0 || src_nrows < 0 || src_row0 + src_nrows > shape_in[1])
        return fail(PXL_EINVAL, "plan_create: source window outside the map");
    if (dst_row0 < 0 || dst_nrows < 0 || dst_row0 + dst_nrows > shape_out[1])
        return fail(PXL_EINVAL, "plan_create: destination window outside the map");

    pxl_reproject_plan* pl = new (std::nothrow) pxl_reproject_plan();
    if (!pl) return fail(PXL_ENOMEM, "plan_create: host allocation failed");
    pl->win = *wcs_in; pl->wout = *wcs_out;
    pl->nx = shape_in[0]; pl->ny = shape_in[1]; pl->nc = shape_in[2];
    pl->src_row0 = src_row0; pl->src_nrows = src_nrows;
    pl->nxo = shape_out[0]; pl->nyo = shape_out[1];
    pl->dst_row0 = dst_row0; pl->dst_nrows = dst_nrows;
    // full-circle test: same 1e-8 threshold as enmap_geom.jl:55
    pl->periodic = fabs((double)pl->nx * fabs(wcs_in->cdelt[0] * wcs_in->unit) - PXL_TWOPI_D) < 1e-8;
    pl->tables_built = false;
    pl->variant = env_int("PXL_REPROJECT_VARIANT", 0);

    hipError_t e = hipGetDevice(&pl->device);
    if (e != hipSuccess) { delete pl; return fail(PXL_ENODEV, "hipGetDevice: %s", hipGetErrorString(e)); }

    // tables: [xfx nxo doubles][yfy nyo doubles][xi0 nxo int32][yj0 nyo int32], 16-B aligned pieces
    size_t nxo = (size_t)pl->nxo, nyo = (size_t)pl->nyo;
    size_t off_xfx = 0;
    size_t off_yfy = off_xfx + ((nxo * 8 + 15) & ~(size_t)15);
    size_t off_xi0 = off_yfy + ((nyo * 8 + 15) & ~(size_t)15);
    size_t off_yj0 = off_xi0 + ((nxo * 4 + 15) & ~(size_t)15);
    size_t off_zero = off_yj0 + ((nyo * 4 + 15) & ~(size_t)15);
    size_t total = off_zero + 64;
    e = hipMalloc(&pl->table_mem, total);
    if (e != hipSuccess) { delete pl; return fail(PXL_ENOMEM, "plan_create: hipMalloc(%zu): %s", total, hipGetErrorString(e)); }
    char* base = (char*)pl->table_mem;
    e = hipMemset(base + off_zero, 0, 64);
    if (e != hipSuccess) { (void)hipFree(pl->table_mem); delete pl; return fail(PXL_EHIP, "plan_create: hipMemset: %s", hipGetErrorString(e)); }
    pl->zero_page = (double*)(base + off_zero);
    pl->xfx = (double*)(base + off_xfx); pl->yfy = (double*)(base + off_yfy);
    pl->xi0 = (int32_t*)(base + off_xi0); pl->yj0 = (int32_t*)(base + off_yj0);

    // host copy of the row cells (identical arithmetic: fmod/div/floor are exact or correctly rounded)
    pl->h_yj0 = new (std::nothrow) int32_t[nyo];
    if (!pl->h_yj0) { (void)hipFree(pl->table_mem); delete pl; return fail(PXL_ENOMEM, "plan_create: host allocation failed"); }
    CarAffine co = car_affine(pl->wout);
    Sky2Pix si = sky2pix_setup(pl->win, pl->nx, pl->ny, 1, PXL_FORM_DIV);
    for (int64_t j = 0; j < pl->nyo; ++j) {
        double fr;
        split_cell(s2p_y(si, p2s_dec(co, (double)(j + 1))), &pl->h_yj0[j], &fr);
    }

    // ---- choose the launch configuration from the RA scale (source columns per output column)
    double sx = fabs((pl->wout.cdelt[0] * pl->wout.unit) / (pl->win.cdelt[0] * pl->win.unit));
    pl->dxpos = ((pl->wout.cdelt[0] * pl->wout.unit) / (pl->win.cdelt[0] * pl->win.unit)) > 0 ? 1 : 0;
    pl->dypos = ((pl->wout.cdelt[1] * pl->wout.unit) / (pl->win.cdelt[1] * pl->win.unit)) > 0 ? 1 : 0;
    double sy = fabs((pl->wout.cdelt[1] * pl->wout.unit) / (pl->win.cdelt[1] * pl->win.unit));
    pl->rh = env_int("PXL_REPROJECT_RH", 32);
    if (pl->rh < 1) pl->rh = 1;
    if (pl->rh > 64) pl->rh = 64;          // one lane per tile row holds the row-table entry
    pl->flags = env_int("PXL_REPROJECT_FLAGS", 0);
    pl->ns = env_int("PXL_REPROJECT_NS", 8);
    if (pl->ns < 4) pl->ns = 4;
    while (pl->ns & (pl->ns - 1)) pl->ns &= pl->ns - 1;       // power of two
    if (pl->ns > 64) pl->ns = 64;
    pl->pf = env_int("PXL_REPROJECT_PF", 3);
    if (pl->pf < 0) pl->pf = 0;
    const int max_seg = PXL_MAXCH * 128;
    auto seg_for = [&](int pairs) -> int64_t {
        // footprint of TW columns: ceil(TW*sx) cells + 2 (tap +1, rounding) + 1 (even alignment) + 2 slack
        double span = ceil((double)(128 * pairs) * sx) + 5.0;
        int64_t s = (int64_t)span;
        return (s + 1) & ~(int64_t)1;
    };
    // lane width: 2 pairs (256 columns per wave) by default; strong up-sampling (>= ~3x) is store-issue bound and
    // gains 15 % from 512 columns per wave (measured on 10800 -> 43200)
    int want = env_int("PXL_REPROJECT_PAIRS", sx <= 0.3 ? 4 : 2);
    if (want != 1 && want != 2 && want != 4) want = 2;
    // LDS-DMA kernel: widest lane width whose slot fits
    pl->pairs_dma = want;
    while (pl->pairs_dma > 1 && seg_for(pl->pairs_dma) > max_seg) pl->pairs_dma >>= 1;
    pl->seg_dma = (int)std::min<int64_t>(seg_for(pl->pairs_dma), max_seg);
    // register-staged kernel: 1 or 2
    pl->pairs = want > 2 ? 2 : want;
    if (seg_for(pl->pairs) > max_seg && pl->pairs == 2) pl->pairs = 1;
    int64_t seg = seg_for(pl->pairs);
    // stageable: the slot fits, never laps the ring of pixels, and rows are not skipped wholesale
    auto stageable = [&](int64_t sg) { return (sg <= max_seg) && !(pl->periodic && sg > pl->nx) && sy <= 3.0; };
    pl->staged_ok = stageable(seg) && stageable(seg_for(pl->pairs_dma));
    pl->seg = (int)(seg <= max_seg ? seg : max_seg);
    pl->vec_load = (pl->nx % 2 == 0);
    *out = pl;
    return PXL_OK;
}

int pxl_reproject_plan_set_variant(pxl_reproject_plan* pl, int variant) {
    if (!pl || variant < 0 || variant > 2) return fail(PXL_EINVAL, "set_variant: bad argument");
    pl->variant = variant;
    return PXL_OK;
}

int pxl_reproject_plan_destroy(pxl_reproject_plan* pl) {
    if (!pl) return PXL_OK;
    if (pl->table_mem) (void)hipFree(pl->table_mem);
    delete[] pl->h_yj0;
    delete pl;
    return PXL_OK;
}

int pxl_reproject_build_tables(pxl_reproject_plan* pl, void* stream) {
    if (!pl) return fail(PXL_EINVAL, "build_tables: null plan");
    CarAffine co = car_affine(pl->wout);
    Sky2Pix si = sky2pix_setup(pl->win, pl->nx, pl->ny, 1, PXL_FORM_DIV);
    hipLaunchKernelGGL(k_build_tables, dim3(stream_grid(pl->nxo + pl->nyo, 256)), dim3(256), 0,
                       (hipStream_t)stream, co, si, pl->nxo, pl->nyo, pl->xi0, pl->xfx, pl->yj0, pl->yfy);
    int rc = check_launch("k_build_tables");
    if (rc == PXL_OK) pl->tables_built = true;
    return rc;
}

int pxl_reproject_execute_rows(pxl_reproject_plan* pl, const double* src, double* dst, int64_t r0, int64_t nr,
                               void* stream) {
    if (!pl) return fail(PXL_EINVAL, "execute: null plan");
    if (r0 < 0 || nr < 0 || r0 + nr > pl->dst_nrows) return fail(PXL_EINVAL, "execute: rows outside the dst window");
    if (nr == 0) return PXL_OK;
    if (!dst || (!src && pl->src_nrows > 0)) return fail(PXL_EINVAL, "execute: null src/dst");
    if (!pl->tables_built) return fail(PXL_EINVAL, "execute_rows: tables not built");
    hipStream_t st = (hipStream_t)stream;

    ReprojParams p;
    memset(&p, 0, sizeof(p));
    p.src = src; p.dst = dst;
    p.xi0 = pl->xi0; p.xfx = pl->xfx; p.yj0 = pl->yj0; p.yfy = pl->yfy;
    p.nx = pl->nx; p.ny = pl->ny; p.src_row0 = pl->src_row0; p.src_nrows = pl->src_nrows;
    p.nxo = pl->nxo; p.dst_row0 = pl->dst_row0; p.dst_nrows = pl->dst_nrows;
    p.r0 = r0; p.nr = nr; p.nc = (int32_t)pl->nc; p.periodic = pl->periodic;

    bool staged = pl->staged_ok;
    if (pl->variant == 1) staged = false;
    if (!staged) {
        int64_t work = ((pl->nxo + 1) / 2) * nr;
        hipLaunchKernelGGL(k_reproject_gather, dim3(stream_grid(work, 256), (unsigned)pl->nc), dim3(256), 0, st, p);
        return check_launch("k_reproject_gather");
    }

    const bool vec = pl->vec_load && (((uintptr_t)src & 15) == 0);
    const bool use_dma = vec && pl->variant != 2;
    const int pairs = use_dma ? pl->pairs_dma : pl->pairs;
    const int TW = 128 * pairs;
    p.seg = use_dma ? pl->seg_dma : pl->seg; p.dxpos = pl->dxpos; p.dypos = pl->dypos; p.flags = pl->flags;
    p.ntx = (int32_t)((pl->nxo + TW - 1) / TW);
    // tile height: the configured rh, halved while the launch would leave the chip short of waves
    // (256 CUs x ~16 resident waves, a few rounds each); small maps and thin strips get shorter tiles
    int rh = pl->rh;
    while (rh > 4 && (int64_t)p.ntx * ((nr + rh - 1) / rh) * pl->nc < 16 * 4096) rh >>= 1;
    p.rh = rh;
    p.nty = (int32_t)((nr + rh - 1) / rh);
    p.ntiles = (int64_t)p.ntx * p.nty * pl->nc;
    p.tiles_per_xcd = (p.ntiles + 7) / 8;
    int64_t nblocks = p.tiles_per_xcd * 8;
    if (nblocks > 0x7fffffffLL) return fail(PXL_EINVAL, "execute: too many tiles (%lld)", (long long)nblocks);
    dim3 grid((unsigned)nblocks), block(64);
    if (use_dma) {
        // LDS-DMA fast path; shrink the ring if it would not fit a CU's LDS comfortably
        p.ns = pl->ns; p.pf = pl->pf; p.zero_page = pl->zero_page;
        while ((size_t)p.ns * p.seg * 8 > 17 * 1024 && p.ns > 4) p.ns >>= 1;    // keep >= 9 waves per CU
        size_t dma_lds = (size_t)p.ns * (size_t)p.seg * sizeof(double);
        return launch_reproject_dma(pairs, (p.seg + 127) / 128, grid, dma_lds, (hipStream_t)stream, p);
    }
    size_t lds_bytes = (size_t)PXL_NS * (size_t)pl->seg * sizeof(double);
    if (pl->pairs == 2) {
        if (vec) hipLaunchKernelGGL((k_reproject_staged<2, true>), grid, block, lds_bytes, st, p);
        else     hipLaunchKernelGGL((k_reproject_staged<2, false>), grid, block, lds_bytes, st, p);
    } else {
        if (vec) hipLaunchKernelGGL((k_reproject_staged<1, true>), grid, block, lds_bytes, st, p);
        else     hipLaunchKernelGGL((k_reproject_staged<1, false>), grid, block, lds_bytes, st, p);
    }
    return check_launch("k_reproject_staged");
}

int pxl_reproject_execute(pxl_reproject_plan* pl, const double* src, double* dst, void* stream) {
    int rc = pxl_reproject_build_tables(pl, stream);
    if (rc) return rc;
    return pxl_reproject_execute_rows(pl, src, dst, 0, pl->dst_nrows, stream);
}

int pxl_reproject_plan_src_rows(const pxl_reproject_plan* pl, int64_t* lo, int64_t* hi) {
    if (!pl || !lo || !hi) return fail(PXL_EINVAL, "plan_src_rows: null argument");
    int64_t l = INT64_MAX, h = INT64_MIN;
    for (int64_t r = 0; r < pl->dst_nrows; ++r) {
        int64_t j0 = pl->h_yj0[pl->dst_row0 + r];
        for (int64_t j = j0; j <= j0 + 1; ++j)
            if (j >= 1 && j <= pl->ny) { if (j - 1 < l) l = j - 1; if (j > h) h = j; }
    }
    if (l > h) { l = 0; h = 0; }
    *lo = l; *hi = h;
    return PXL_OK;
}

int pxl_reproject_plan_rows_covered(const pxl_reproject_plan* pl, int64_t have_lo, int64_t have_hi, int64_t* lo,
                                    int64_t* hi) {
    if (!pl || !lo || !hi) return fail(PXL_EINVAL, "plan_rows_covered: null argument");
    // longest run of output rows (relative) whose in-map taps j0, j0+1 all lie in [have_lo, have_hi)
    int64_t best_lo = 0, best_hi = 0, cur_lo = -1;
    for (int64_t r = 0; r <= pl->dst_nrows; ++r) {
        bool ok = false;
        if (r < pl->dst_nrows) {
            int64_t j0 = pl->h_yj0[pl->dst_row0 + r];
            ok = true;
            for (int64_t j = j0; j <= j0 + 1; ++j)
                if (j >= 1 && j <= pl->ny && !(j - 1 >= have_lo && j - 1 < have_hi)) ok = false;
        }
        if (ok) { if (cur_lo < 0) cur_lo = r; }
        else if (cur_lo >= 0) {
            if (r - cur_lo > best_hi - best_lo) { best_lo = cur_lo; best_hi = r; }
            cur_lo = -1;
        }
    }
    *lo = best_lo; *hi = best_hi;
    return PXL_OK;
}

int pxl_reproject_car_bilinear_f64(const pxl_car_wcs* wcs_in, const int64_t shape_in[3], const double* src,
                                   const pxl_car_wcs* wcs_out, const int64_t shape_out[2], double* dst,
                                   void* stream) {
    if (!shape_in || !shape_out) return fail(PXL_EINVAL, "reproject: null shape");
    pxl_reproject_plan* pl = nullptr;
    int rc = pxl_reproject_plan_create(wcs_in, shape_in, 0, shape_in[1], wcs_out, shape_out, 0, shape_out[1], &pl);
    if (rc) return rc;
    rc = pxl_reproject_execute(pl, src, dst, stream);
    if (rc == PXL_OK) {
        hipError_t e = hipStreamSynchronize((hipStream_t)stream);
        if (e != hipSuccess) rc = fail(PXL_EHIP, "reproject: %s", hipGetErrorString(e));
    }
    pxl_reproject_plan_destroy(pl);
    return rc;
}

int pxl_reproject_generic_bilinear_f64(const pxl_car_wcs* wcs_in, int proj_in, const int64_t shape_in[3],
                                       const double* src, const pxl_car_wcs* wcs_out, int proj_out,
                                       const int64_t shape_out[2], double* dst, void* stream) {
    if (!wcs_ok(wcs_in) || !wcs_ok(wcs_out)) return fail(PXL_EINVAL, "reproject_generic: invalid WCS");
    if (!shape_in || !shape_out) return fail(PXL_EINVAL, "reproject_generic: null shape");
    if (shape_in[0] < 1 || shape_in[1] < 1 || shape_in[2] < 1 || shape_out[0] < 1 || shape_out[1] < 1)
        return fail(PXL_EINVAL, "reproject_generic: shapes must be positive");
    if ((proj_in != PXL_PROJ_CAR && proj_in != PXL_PROJ_TAN) || (proj_out != PXL_PROJ_CAR && proj_out != PXL_PROJ_TAN))
        return fail(PXL_EINVAL, "reproject_generic: unknown projection code");
    if (!src || !dst) return fail(PXL_EINVAL, "reproject_generic: null src/dst");
    GenericParams p;
    memset(&p, 0, sizeof(p));
    p.src = src; p.dst = dst;
    p.nx = shape_in[0]; p.ny = shape_in[1]; p.nc = (int32_t)shape_in[2];
    p.nxo = shape_out[0]; p.nyo = shape_out[1];
    p.proj_in = proj_in; p.proj_out = proj_out;
    p.periodic = (proj_in == PXL_PROJ_CAR) &&
                 fabs((double)p.nx * fabs(wcs_in->cdelt[0] * wcs_in->unit) - PXL_TWOPI_D) < 1e-8;
    if (proj_out == PXL_PROJ_TAN) p.out_tan = tan_setup(*wcs_out); else p.out_car = car_affine(*wcs_out);
    if (proj_in == PXL_PROJ_TAN) p.in_tan = tan_setup(*wcs_in);
    else p.in_car = sky2pix_setup(*wcs_in, p.nx, p.ny, 1, PXL_FORM_DIV);
    hipLaunchKernelGGL(k_reproject_generic, dim3(stream_grid(p.nxo * p.nyo, 256)), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("k_reproject_generic");
}

int pxl_sample_car_bilinear_f64(const pxl_car_wcs* wcs_in, const int64_t shape_in[3], const double* src,
                                int64_t src_row0, int64_t src_nrows, int64_t n, const double* sky, double* out,
                                void* stream) {
    if (!wcs_ok(wcs_in) || !shape_in) return fail(PXL_EINVAL, "sample: invalid WCS/shape");
    if (shape_in[0] < 1 || shape_in[1] < 1 || shape_in[2] < 1) return fail(PXL_EINVAL, "sample: shapes must be positive");
    if (src_row0 < 0 || src_nrows < 0 || src_row0 + src_nrows > shape_in[1])
        return fail(PXL_EINVAL, "sample: source window outside the map");
    if (n < 0 || (n > 0 && (!sky || !out || (!src && src_nrows > 0)))) return fail(PXL_EINVAL, "sample: null buffer or negative n");
    if (((uintptr_t)sky & 15) != 0) return fail(PXL_EINVAL, "sample: 2xN buffer must be 16-byte aligned");
    if (n == 0) return PXL_OK;
    Sky2Pix s = sky2pix_setup(*wcs_in, shape_in[0], shape_in[1], 1, PXL_FORM_RECIP);
    int periodic = fabs((double)shape_in[0] * fabs(wcs_in->cdelt[0] * wcs_in->unit) - PXL_TWOPI_D) < 1e-8;
    hipLaunchKernelGGL(k_sample_bilinear, dim3(stream_grid((n + PXL_SUNR - 1) / PXL_SUNR, 256)), dim3(256), 0,
                       (hipStream_t)stream, s, src, shape_in[0], shape_in[1], (int32_t)shape_in[2], src_row0, src_nrows,
                       periodic, n, (const double2*)sky, out);
    return check_launch("k_sample_bilinear");
}

int pxl_fits_decode_f64(const void* raw_be, double* dst, int64_t n, int bitpix, void* stream) {
    if (n < 0 || (n > 0 && (!raw_be || !dst))) return fail(PXL_EINVAL, "fits_decode: null buffer or negative n");
    if (bitpix != -64 && bitpix != -32) return fail(PXL_EINVAL, "fits_decode: BITPIX %d not supported (only -64, -32)", bitpix);
    if (n == 0) return PXL_OK;
    hipLaunchKernelGGL(k_bswap_to_f64, dim3(stream_grid((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, raw_be, dst, n, bitpix);
    return check_launch("k_bswap_to_f64");
}

int pxl_fits_encode_f64(const double* src, void* raw_be, int64_t n, void* stream) {
    if (n < 0 || (n > 0 && (!raw_be || !src))) return fail(PXL_EINVAL, "fits_encode: null buffer or negative n");
    if (n == 0) return PXL_OK;
    hipLaunchKernelGGL(k_f64_to_be, dim3(stream_grid((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, src, (uint64_t*)raw_be, n);
    return check_launch("k_f64_to_be");
}

int pxl_fill_random_f64(double* dst, int64_t n, uint64_t seed, uint64_t offset, int kind, void* stream) {
    if (n < 0 || (n > 0 && !dst)) return fail(PXL_EINVAL, "fill_random: null buffer or negative n");
    if (n == 0) return PXL_OK;
    hipLaunchKernelGGL(k_fill_random, dim3(stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, dst, n, seed,
                       offset, kind);
    return check_launch("k_fill_random");
}

int pxl_fill_sphere_points_f64(double* sky, int64_t n, uint64_t seed, uint64_t offset, void* stream) {
    if (n < 0 || (n > 0 && !sky)) return fail(PXL_EINVAL, "fill_sphere: null buffer or negative n");
    if (((uintptr_t)sky & 15) != 0) return fail(PXL_EINVAL, "fill_sphere: 2xN buffer must be 16-byte aligned");
    if (n == 0) return PXL_OK;
    hipLaunchKernelGGL(k_fill_sphere, dim3(stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, (double2*)sky,
                       n, seed, offset);
    return check_launch("k_fill_sphere");
}

}  // extern "C"
