// pxl_sample.h -- non-separable paths: CAR<->Gnomonic reprojection and the scattered bilinear sampler; included by pxl_kernels.hip (one translation unit, -ffp-contract=off).
#pragma once

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
#ifndef PXL_SUNR
#define PXL_SUNR 4
#endif
__device__ inline int64_t wrap_col(int64_t i, int64_t nx) {          // 1-based column of a periodic map
    if (i >= 1 - nx && i <= 2 * nx) { if (i > nx) i -= nx; else if (i < 1) i += nx; return i; }
    i = (i - 1) % nx; if (i < 0) i += nx; return i + 1;
}
template <typename T>
__global__ __launch_bounds__(256) void k_sample_bilinear(Sky2Pix s, const T* __restrict__ src, int64_t nx,
                                                         int64_t ny, int32_t nc, int64_t row0, int64_t nrows,
                                                         int periodic, int64_t n, const double2* __restrict__ sky,
                                                         T* __restrict__ out) {
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
            const T* pl = src + (int64_t)c * plane;
            double m00[PXL_SUNR], m10[PXL_SUNR], m01[PXL_SUNR], m11[PXL_SUNR];
#pragma unroll
            for (int u = 0; u < PXL_SUNR; ++u) {
                m00[u] = o00[u] >= 0 ? (double)pl[o00[u]] : 0.0;
                m10[u] = o10[u] >= 0 ? (double)pl[o10[u]] : 0.0;
                m01[u] = o01[u] >= 0 ? (double)pl[o01[u]] : 0.0;
                m11[u] = o11[u] >= 0 ? (double)pl[o11[u]] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < PXL_SUNR; ++u) {
                int64_t k = k0 + u * blockDim.x;
                double top = (1 - fx[u]) * m00[u] + fx[u] * m10[u];
                double bot = (1 - fx[u]) * m01[u] + fx[u] * m11[u];
                double v = (1 - fy[u]) * top + fy[u] * bot;
                if (k < n) out[(int64_t)c * n + k] = (T)(fin[u] ? v : __builtin_nan(""));
            }
        }
    }
}

// ---- row-pair layout for scattered sampling.  The gather above is bound by random 64-byte sector fetches (about
// 2.25 per point: the two taps of a row are neighbours, the two rows are nx elements apart).  In the pair layout
// element (p, i) holds (v[p-1][i], v[p][i]) -- rows outside the resident window read as zero, p = 0 .. nrows -- so
// the whole 2x2 neighbourhood of a point is two ADJACENT 2-element entries: about 1.25 sectors per point, for
// twice the map's footprint (288 GB of HBM is there to be used) and one streaming pass to build it.  Same taps,
// same arithmetic: results are bit-identical to k_sample_bilinear.
template <typename T> struct Vec2T;
template <> struct Vec2T<double> { typedef double2 type; };
template <> struct Vec2T<float>  { typedef float2 type; };

template <typename T>
__global__ __launch_bounds__(256) void k_build_rowpairs(const T* __restrict__ src, int64_t nx, int64_t nrows,
                                                        typename Vec2T<T>::type* __restrict__ pairs) {
    // block = (256 columns, PXL_POS_ROWS pair rows, component): a lane walks down its column carrying the row above
    typedef typename Vec2T<T>::type T2;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nx) return;
    const int64_t c = blockIdx.z;
    const T* pl = src + c * nx * nrows;
    T2* out = pairs + c * nx * (nrows + 1);
    const int64_t p0 = (int64_t)blockIdx.y * PXL_POS_ROWS;
    const int64_t p1 = (p0 + PXL_POS_ROWS <= nrows) ? p0 + PXL_POS_ROWS : nrows + 1;     // pair rows [p0, p1)
    T above = (p0 >= 1) ? pl[(p0 - 1) * nx + i] : (T)0;
    for (int64_t p = p0; p < p1; ++p) {
        const T here = (p < nrows) ? pl[p * nx + i] : (T)0;
        T2 v; v.x = above; v.y = here;
        out[p * nx + i] = v;
        above = here;
    }
}

// points per lane per trip of k_sample_pairs: Float32 entries (8 B) run 17 % faster with 2 than with 4 on the
// 0.5-arcmin map (29.5 vs 25.2 Gpts/s, same-box A/B); Float64 is indifferent
template <typename T> struct PairsUnroll { static constexpr int value = PXL_SUNR; };
template <> struct PairsUnroll<float> { static constexpr int value = 2; };
template <typename T>
__global__ __launch_bounds__(256) void k_sample_pairs(Sky2Pix s, const typename Vec2T<T>::type* __restrict__ pairs,
                                                      int64_t nx, int64_t ny, int32_t nc, int64_t row0, int64_t nrows,
                                                      int periodic, int64_t n, const double2* __restrict__ sky,
                                                      T* __restrict__ out) {
    typedef typename Vec2T<T>::type T2;
    constexpr int SUNR = PairsUnroll<T>::value;
    const int64_t chunk = (int64_t)blockDim.x * SUNR;
    const int64_t plane = nx * (nrows + 1);
    for (int64_t k0 = (int64_t)blockIdx.x * chunk + threadIdx.x; k0 < n; k0 += (int64_t)gridDim.x * chunk) {
        double2 ad[SUNR];
#pragma unroll
        for (int u = 0; u < SUNR; ++u) {
            int64_t k = k0 + u * blockDim.x;
            ad[u] = (k < n) ? sky[k] : make_double2(0.0, 0.0);
        }
        int64_t oa[SUNR], ob[SUNR];                  // pair-entry offsets of columns i0 and i0 + 1, -1 = zeros
        double fx[SUNR], fy[SUNR];
        bool fin[SUNR];
#pragma unroll
        for (int u = 0; u < SUNR; ++u) {
            double x = s2p_x(s, ad[u].x), y = s2p_y(s, ad[u].y);
            fin[u] = isfinite(x) && isfinite(y);
            int32_t i0, j0;
            split_cell(x, &i0, &fx[u]);
            split_cell(y, &j0, &fy[u]);
            int64_t ia = i0, ib = (int64_t)i0 + 1;
            bool oka = true, okb = true;
            if (periodic) { ia = wrap_col(ia, nx); ib = wrap_col(ib, nx); }
            else { oka = (ia >= 1 && ia <= nx); okb = (ib >= 1 && ib <= nx); }
            const int64_t p = (int64_t)j0 - row0;            // entry p holds resident rows p-1 (cell row j0) and p
            const bool rows = (p >= 0 && p <= nrows);
            oa[u] = (rows && oka) ? p * nx + (ia - 1) : -1;
            ob[u] = (rows && okb) ? p * nx + (ib - 1) : -1;
        }
        for (int c = 0; c < nc; ++c) {
            const T2* pl = pairs + (int64_t)c * plane;
            T2 a[SUNR], b[SUNR];
#pragma unroll
            for (int u = 0; u < SUNR; ++u) {
                T2 z; z.x = (T)0; z.y = (T)0;
                a[u] = oa[u] >= 0 ? pl[oa[u]] : z;
                b[u] = ob[u] >= 0 ? pl[ob[u]] : z;
            }
#pragma unroll
            for (int u = 0; u < SUNR; ++u) {
                int64_t k = k0 + u * blockDim.x;
                double top = (1 - fx[u]) * (double)a[u].x + fx[u] * (double)b[u].x;
                double bot = (1 - fx[u]) * (double)a[u].y + fx[u] * (double)b[u].y;
                double v = (1 - fy[u]) * top + fy[u] * bot;
                if (k < n) out[(int64_t)c * n + k] = (T)(fin[u] ? v : __builtin_nan(""));
            }
        }
    }
}
