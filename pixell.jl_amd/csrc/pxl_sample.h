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
#define PXL_SUNR 4
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
