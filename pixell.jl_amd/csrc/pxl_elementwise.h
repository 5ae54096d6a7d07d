// pxl_elementwise.h -- streaming evaluators: pix2sky / sky2pix on 2xN and SoA batches; included by pxl_kernels.hip (one translation unit, -ffp-contract=off).
#pragma once

// ------------------------------------------------------------------------------------------------
// elementwise evaluators (A9-A13): one (c1, c2) pair = 16 B in, 16 B out per lane
// ------------------------------------------------------------------------------------------------
// Each lane handles UNR points per trip, all loads issued before the arithmetic.  With one contiguous chunk per block
// the affine-only forms are fastest at UNR = 1 (2.10 vs 2.27 ms per 4e8 points, same-box A/B: 76 % of the HBM peak,
// the mixed read/write ceiling of the part); the forms that rewind (two exact fmod per point) prefer 2.
template <int PXL_UNR>
__global__ __launch_bounds__(256) void k_pix2sky_pairs(CarAffine c, int64_t n, const double2* pix,
                                                       double2* sky, int mode, const int32_t* gate) {
    if (gate && *gate == 0) return;      // fallback launches of the unwind path: run only when the fused form failed
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

// SoA forms: two adjacent points per lane so that every access is 16 B per lane (1 KiB per wave instruction); `vec`
// = all four arrays 16-byte aligned (otherwise, and for an odd last point, scalar accesses).
__global__ __launch_bounds__(256) void k_pix2sky_soa(CarAffine c, int64_t n, const double* __restrict__ ip,
                                                     const double* __restrict__ jp, double* __restrict__ ra,
                                                     double* __restrict__ dec, int safe, int vec) {
    const int64_t npair = (n + 1) / 2;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < npair; t += stride) {
        const int64_t k = 2 * t;
        const bool two = k + 1 < n;
        double2 pi, pj;
        if (vec && two) { pi = *reinterpret_cast<const double2*>(ip + k); pj = *reinterpret_cast<const double2*>(jp + k); }
        else { pi.x = ip[k]; pj.x = jp[k]; pi.y = two ? ip[k + 1] : 0.0; pj.y = two ? jp[k + 1] : 0.0; }
        double a0 = p2s_ra(c, pi.x), d0 = p2s_dec(c, pj.x), a1 = p2s_ra(c, pi.y), d1 = p2s_dec(c, pj.y);
        if (safe) {
            a0 = rewind(a0, PXL_TWOPI_D, 0.0); d0 = rewind(d0, PXL_TWOPI_D, 0.0);
            a1 = rewind(a1, PXL_TWOPI_D, 0.0); d1 = rewind(d1, PXL_TWOPI_D, 0.0);
        }
        if (vec && two) { *reinterpret_cast<double2*>(ra + k) = make_double2(a0, a1); *reinterpret_cast<double2*>(dec + k) = make_double2(d0, d1); }
        else { ra[k] = a0; dec[k] = d0; if (two) { ra[k + 1] = a1; dec[k + 1] = d1; } }
    }
}

template <int PXL_UNR>
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
                                                     double* __restrict__ jp, int vec) {
    const int64_t npair = (n + 1) / 2;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < npair; t += stride) {
        const int64_t k = 2 * t;
        const bool two = k + 1 < n;
        double2 a, d;
        if (vec && two) { a = *reinterpret_cast<const double2*>(ra + k); d = *reinterpret_cast<const double2*>(dec + k); }
        else { a.x = ra[k]; d.x = dec[k]; a.y = two ? ra[k + 1] : 0.0; d.y = two ? dec[k + 1] : 0.0; }
        const double x0 = s2p_x(s, a.x), y0 = s2p_y(s, d.x), x1 = s2p_x(s, a.y), y1 = s2p_y(s, d.y);
        if (vec && two) { *reinterpret_cast<double2*>(ip + k) = make_double2(x0, x1); *reinterpret_cast<double2*>(jp + k) = make_double2(y0, y1); }
        else { ip[k] = x0; jp[k] = y0; if (two) { ip[k + 1] = x1; jp[k + 1] = y1; } }
    }
}
