// pxl_unwrap.h -- unwind! as a verified integer scan, the exact serial fallback, and rewind!; included by pxl_kernels.hip (one translation unit, -ffp-contract=off).
#pragma once

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
