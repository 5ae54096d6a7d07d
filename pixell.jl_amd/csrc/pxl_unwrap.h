// pxl_unwrap.h -- unwind! as a verified integer scan, the exact serial fallback, and rewind!; included by pxl_kernels.hip (one translation unit, -ffp-contract=off).
#pragma once
#include <type_traits>

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
                                                     int8_t* __restrict__ c, const int32_t* __restrict__ gate) {
    // m2: nrow x N interleaved rewound values (nrow = 2 for coordinate batches, 1 for a plain vector); c: [nrow][n]
    if (gate && *gate == 0) return;
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
    for (int64_t blk = blockIdx.x; blk < nb; blk += gridDim.x) {
    const int64_t base = blk * PXL_SCAN_BLOCK + (int64_t)threadIdx.x * PXL_SCAN_ITEMS;
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
    if (threadIdx.x == 255) bsum[row * nb + blk] = woff + incl;
    __syncthreads();
    }
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
                                                      int64_t nb, const int32_t* __restrict__ flag,
                                                      const int32_t* __restrict__ gate) {
    // flag[0]: pass 1 found mismatches; flag[1]: pass 2 (run only then) still found some
    if (gate && *gate == 0) return;         // the fused path already produced the answer
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

// ------------------------------------------------------------------------------------------------
// Fused fast path of unwind!: the same verified integer scan, but the rewound values m never touch memory.
//   k_unwind_sums   reads the input once, forms m in registers, sums the nominal increments of each 64*U-point
//                   wave chunk, remembers the m just before each chunk and the first NaN of each coordinate row
//                   (as the maximum of ~index, so that zero-filled scratch means "none")
//   k_scan_wsums    exclusive scan of the chunk sums (one block)
//   k_unwind_apply  reads the input again, re-forms m and the increments, scans them inside the wave, checks every
//                   element against the reference's floating-point recurrence and writes y = m - r*P + ref
// 48 B per 2xN point instead of ~124 B for the multi-pass form above, which stays as the fallback when the
// check fails (steps within rounding of half a period).  NaN/Inf inputs are exact here: the recurrence turns
// every element from the first NaN of a row onwards into NaN, which is what k_unwind_apply writes.
// In-place calls verify first (no store) and store in a second, device-gated launch, so the input survives for
// the fallback.
// ------------------------------------------------------------------------------------------------
#ifndef PXL_UW_G
#define PXL_UW_G 4          // 16-byte loads in flight per lane
#endif

struct UwSrcPix2 {          // m = rewind(pix2sky affine) - ref, ref = 0   (pix2sky!(...; safe=true), car_proj.jl:104-112)
    static constexpr int NROW = 2;
    typedef double2 raw_t;
    CarAffine c; const double2* p; double period, ref, rperiod;
    __device__ inline raw_t load(int64_t k) const { return p[k]; }
    __device__ inline raw_t zero() const { return make_double2(0.0, 0.0); }
    __device__ inline void to_m(raw_t v, double* m) const {
        const double a = p2s_ra(c, v.x), d = p2s_dec(c, v.y);
        // Coordinates already inside [-P/2, P/2) -- every pixel of a full-sky map, every point of a patch: the wave votes.  For
        // 0 <= x < P the C fmod(x, P) IS x and Julia's mod adds nothing, so rewind is (0 + x) - P/2 with x = (a - 0) + P/2: two
        // additions instead of the ~22 instructions of the exact quotient machinery, the same bits (rewind0_try walks the same
        // values: q = 0 or corrected to 0, r = fma(-0, P, x) = x, flag 0).
        const double half = PXL_TWOPI_D / 2;
        const double x0 = (a - 0.0) + half, x1 = (d - 0.0) + half;
        if (__builtin_expect(__all(x0 >= 0.0 && x0 < PXL_TWOPI_D && x1 >= 0.0 && x1 < PXL_TWOPI_D), 1)) {
            m[0] = (x0 - half) - 0.0;
            m[1] = (x1 - half) - 0.0;
            return;
        }
        bool ok0, ok1;
        m[0] = rewind0_try(a, PXL_TWOPI_D, rperiod, &ok0) - 0.0;
        m[1] = rewind0_try(d, PXL_TWOPI_D, rperiod, &ok1) - 0.0;
        if (__builtin_expect(!(ok0 && ok1), 0)) {                   // NaN/Inf or a huge quotient: library fmod
            m[0] = rewind(a, PXL_TWOPI_D, 0.0, rperiod) - 0.0;
            m[1] = rewind(d, PXL_TWOPI_D, 0.0, rperiod) - 0.0;
        }
    }
    static __device__ inline void store(raw_t* out, int64_t k, const double* y) { out[k] = make_double2(y[0], y[1]); }
};
struct UwSrcAng2 {          // unwind!(angles2xN; dims=2): m = rewind(a) - ref   (enmap_ops.jl:26-28)
    static constexpr int NROW = 2;
    typedef double2 raw_t;
    const double2* p; double period, ref, rperiod;
    __device__ inline raw_t load(int64_t k) const { return p[k]; }
    __device__ inline raw_t zero() const { return make_double2(0.0, 0.0); }
    __device__ inline void to_m(raw_t v, double* m) const {
        // the same vote as above for angles already within half a period of ref: rewind = (ref + x) - P/2 with x = (a - ref) + P/2
        const double half = period / 2;
        const double x0 = (v.x - ref) + half, x1 = (v.y - ref) + half;
        if (__builtin_expect(__all(x0 >= 0.0 && x0 < period && x1 >= 0.0 && x1 < period) && period >= 0x1p-900 && period <= 0x1p900, 1)) {
            m[0] = ((ref + x0) - half) - ref;
            m[1] = ((ref + x1) - half) - ref;
            return;
        }
        bool ok0, ok1;
        m[0] = rewind_try(v.x, period, ref, rperiod, &ok0) - ref;
        m[1] = rewind_try(v.y, period, ref, rperiod, &ok1) - ref;
        if (__builtin_expect(!(ok0 && ok1), 0)) {
            m[0] = rewind(v.x, period, ref, rperiod) - ref;
            m[1] = rewind(v.y, period, ref, rperiod) - ref;
        }
    }
    static __device__ inline void store(raw_t* out, int64_t k, const double* y) { out[k] = make_double2(y[0], y[1]); }
};
struct UwSrcAng1 {          // the same on a plain vector
    static constexpr int NROW = 1;
    typedef double raw_t;
    const double* p; double period, ref, rperiod;
    __device__ inline raw_t load(int64_t k) const { return p[k]; }
    __device__ inline raw_t zero() const { return 0.0; }
    __device__ inline void to_m(raw_t v, double* m) const { m[0] = rewind(v, period, ref, rperiod) - ref; m[1] = 0.0; }
    static __device__ inline void store(raw_t* out, int64_t k, const double* y) { out[k] = y[0]; }
};

// wave-level data movement on the VALU (DPP / readlane) instead of ds_bpermute shuffles
// lane l gets lane l-1's value, lane 0 gets `first` (wave_shr:1 leaves lane 0's destination -- the `old` operand -- alone)
__device__ inline double uw_shr1_first(double v, double first) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(__double2loint(first), lo, 0x138, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(__double2hiint(first), hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ inline double uw_lane63(double v) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
__device__ inline int uw_scan64(int v) {                   // inclusive sum over the 64 lanes
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);        // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);        // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);        // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);        // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);       // row_bcast:15 into rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);       // row_bcast:31 into rows 2, 3
    return v;
}

__device__ inline int uw_wave_total(int v) { return __builtin_amdgcn_readlane(uw_scan64(v), 63); }

// m, the m of the element before (mp) and the nominal increment c of this lane's element.  The increment is only a
// guess (a multiply by ~1/P instead of the reference's division): k_unwind_apply checks every element with the
// reference's own formula, and both kernels form the guess identically.
template <class SRC>
__device__ inline void uw_element(const SRC& s, int lane, int64_t k, bool valid, typename SRC::raw_t raw,
                                  const double* mlast, double* m, double* mp, int* c) {
    s.to_m(raw, m);
#pragma unroll
    for (int r = 0; r < SRC::NROW; ++r) {
        mp[r] = uw_shr1_first(m[r], mlast[r]);
        // round((m - mp) / P) for rewound values (|m - mp| < P) is +1 above half a period, -1 below minus half, else 0 (a tie rounds
        // to the even 0): two compares instead of multiply, clamp, rint and convert.  Always within {-1, 0, 1}, so the 16-bit scan
        // fields stay consistent whatever the input; a NaN gives 0 -- every element from the first NaN of a row on is written as
        // NaN whatever its count.
        const double d = m[r] - mp[r], half = s.period / 2;
        const int g = (d > half ? 1 : 0) - (d < -half ? 1 : 0);
        c[r] = (valid && k > 0) ? g : 0;
    }
}

template <class SRC>
__global__ __launch_bounds__(64) void k_unwind_sums(SRC src, int64_t n, int U, int2* __restrict__ wsum,
                                                    double2* __restrict__ wprev, unsigned long long* __restrict__ firstnan) {
    const int lane = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * 64 * U;
    double mlast[2];
    src.to_m(base > 0 ? src.load(base - 1) : src.zero(), mlast);
    if (lane == 0) wprev[blockIdx.x] = make_double2(mlast[0], mlast[1]);
    int sum[2] = {0, 0};
    unsigned long long nanat[2] = {~0ULL, ~0ULL};
    for (int u0 = 0; u0 < U; u0 += PXL_UW_G) {
        typename SRC::raw_t v[PXL_UW_G];
#pragma unroll
        for (int g = 0; g < PXL_UW_G; ++g) {
            const int64_t k = base + (int64_t)(u0 + g) * 64 + lane;
            v[g] = (k < n) ? src.load(k) : src.zero();
        }
#pragma unroll
        for (int g = 0; g < PXL_UW_G; ++g) {
            const int64_t k = base + (int64_t)(u0 + g) * 64 + lane;
            double m[2], mp[2];
            int c[2];
            uw_element(src, lane, k, k < n, v[g], mlast, m, mp, c);
#pragma unroll
            for (int r = 0; r < SRC::NROW; ++r) {
                sum[r] += c[r];
                if (k < n && m[r] != m[r] && (unsigned long long)k < nanat[r]) nanat[r] = (unsigned long long)k;
                mlast[r] = uw_lane63(m[r]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < SRC::NROW; ++r) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            sum[r] += __shfl_xor(sum[r], off, 64);
            const unsigned long long o = __shfl_xor(nanat[r], off, 64);
            nanat[r] = o < nanat[r] ? o : nanat[r];
        }
    }
    if (lane == 0) {
        wsum[blockIdx.x] = make_int2(sum[0], sum[1]);
#pragma unroll
        for (int r = 0; r < SRC::NROW; ++r)
            if (nanat[r] != ~0ULL) atomicMax(&firstnan[r], ~nanat[r]);     // stored complemented: zero = no NaN
    }
}

// exclusive scan of the chunk sums, in place: one block sweeps tiles of 4096 entries (4 consecutive per thread)
__global__ __launch_bounds__(1024) void k_scan_wsums(int64_t nw, int2* __restrict__ wsum) {
    __shared__ int2 part[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int2 carry = make_int2(0, 0);
    for (int64_t t0 = 0; t0 < nw; t0 += 4096) {
        const int64_t b = t0 + 4 * (int64_t)threadIdx.x;
        int2 v[4];
        if (b + 3 < nw) {
            const int4 q0 = *reinterpret_cast<const int4*>(wsum + b), q1 = *reinterpret_cast<const int4*>(wsum + b + 2);
            v[0] = make_int2(q0.x, q0.y); v[1] = make_int2(q0.z, q0.w); v[2] = make_int2(q1.x, q1.y); v[3] = make_int2(q1.z, q1.w);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = (b + i < nw) ? wsum[b + i] : make_int2(0, 0);
        }
        const int2 t = make_int2(v[0].x + v[1].x + v[2].x + v[3].x, v[0].y + v[1].y + v[2].y + v[3].y);
        const int2 incl = make_int2(uw_scan64(t.x), uw_scan64(t.y));
        if (lane == 63) part[wave] = incl;
        __syncthreads();
        int2 run = make_int2(carry.x + incl.x - t.x, carry.y + incl.y - t.y);
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const int2 pw = part[w];
            if (w < wave) { run.x += pw.x; run.y += pw.y; }
            carry.x += pw.x; carry.y += pw.y;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (b + i < nw) wsum[b + i] = run;
            run.x += v[i].x; run.y += v[i].y;
        }
        __syncthreads();
    }
}

template <class SRC, bool WRITE>
__global__ __launch_bounds__(64) void k_unwind_apply(SRC src, typename SRC::raw_t* out, int64_t n, int U,
                                                     const int2* __restrict__ woff, const double2* __restrict__ wprev,
                                                     const unsigned long long* __restrict__ firstnan,
                                                     int32_t* __restrict__ flag, const int32_t* __restrict__ skip_if_set) {
    if (skip_if_set && *skip_if_set) return;
    const int lane = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * 64 * U;
    const double P = src.period, rP = src.rperiod, ref = src.ref;
    const double2 wp = wprev[blockIdx.x];
    double mlast[2] = {wp.x, wp.y};
    const int2 w0 = woff[blockIdx.x];
    int carry[2] = {w0.x, w0.y};                       // r of the element just before this chunk
    const unsigned long long fn[2] = {~firstnan[0], ~firstnan[1]};        // index of the first NaN of each row
    bool bad = false;
    for (int u0 = 0; u0 < U; u0 += PXL_UW_G) {
        typename SRC::raw_t v[PXL_UW_G];
#pragma unroll
        for (int g = 0; g < PXL_UW_G; ++g) {
            const int64_t k = base + (int64_t)(u0 + g) * 64 + lane;
            v[g] = (k < n) ? src.load(k) : src.zero();
        }
#pragma unroll
        for (int g = 0; g < PXL_UW_G; ++g) {
            const int64_t k = base + (int64_t)(u0 + g) * 64 + lane;
            const bool valid = k < n;
            double m[2], mp[2], y[2];
            int c[2];
            uw_element(src, lane, k, valid, v[g], mlast, m, mp, c);
            // inclusive wave scan of both rows at once: (c + 1) in 16-bit fields, sums <= 128
            int s = (c[0] + 1) | (SRC::NROW == 2 ? (c[1] + 1) << 16 : 0);
            s = uw_scan64(s);
            const int tot = __builtin_amdgcn_readlane(s, 63);
#pragma unroll
            for (int r = 0; r < SRC::NROW; ++r) {
                const int field = r == 0 ? (s & 0xffff) : (s >> 16);
                const int rr = carry[r] + field - (lane + 1);            // r_k
                carry[r] += (r == 0 ? (tot & 0xffff) : (tot >> 16)) - 64;
                mlast[r] = uw_lane63(m[r]);
                if (!valid) continue;
                if ((unsigned long long)k >= fn[r]) { y[r] = __builtin_nan("") + ref; continue; }
                if (k > 0) {
                    const double yprev = mp[r] - (double)(rr - c[r]) * P;   // y[k-1] as the reference forms it
                    const double a = m[r] - yprev;
                    const double qa = a * rP;
                    // rint(a / P) == r_k is certain when the approximate quotient is well inside (r-1/2, r+1/2): |r_k| < 2^31, so
                    // qa is off by less than 2^31 * 4e-16 = 1e-6 and a fixed margin of 1e-4 is safe; otherwise (ties, mismatches:
                    // two elements in ten thousand) divide exactly like the reference
                    if (!(fabs(qa - (double)rr) < 0.4999)) {
                        const double q = a / P;
                        if (!(rint(q) == (double)rr)) bad = true;
                    }
                }
                y[r] = (m[r] - (double)(k > 0 ? rr : 0) * P) + ref;         // k = 0: m - 0 = m, bit for bit
            }
            if (WRITE && valid) SRC::store(out, k, y);
        }
    }
    if (__any(bad) && lane == 0) atomicOr(flag, 1);
}

// ------------------------------------------------------------------------------------------------
// ONE-PASS form (round 4): the same verified integer scan with the carry between wave chunks handed over INSIDE the launch
// (decoupled look-back), so the input is read once and the exact rewind is evaluated once: 32 B per 2xN point and ~125 VALU
// instructions instead of 48 B and ~215 for k_unwind_sums + k_unwind_apply (profiles/r04_unwind_counters.txt).  Out-of-place
// calls only: a chunk needs the element just before it as the caller gave it, and the fallback needs the whole input untouched.
//
// One workgroup per chunk (PXL_UW1_WAVES waves of 64 * (PXL_UW1_U + PXL_UW1_UL) points each); the rewound values stay in registers and LDS between the sum and the apply step.
//   1. chunk id = a ticket (atomic counter): every chunk with a smaller id has STARTED, and a started chunk publishes its
//      aggregate before it waits for anything -- no wave ever waits for a wave that has not been dispatched
//   2. m, the nominal increments, their sum T (per coordinate row) and "a NaN in this chunk"       -> link.agg
//   3. look-back: lanes read the 64 links before the chunk; up to the nearest one that already carries an inclusive prefix they
//      add aggregates; if an aggregate is still missing the wave polls again; a window without a prefix moves 64 links further
//   4. inclusive prefix = exclusive + T                                                           -> link.pre0, link.pre1
//   5. r_k, the check of every element against the reference's recurrence, y = m - r P + ref: the arithmetic of k_unwind_apply
// A link field is ONE naturally aligned 8-byte {payload, tag} written by one agent-scope relaxed store (global_store_dwordx2 sc1)
// and read by agent-scope relaxed loads (sc1: served by the L2, never by a stale L1): data and "ready" arrive together, so no
// fence and no ordering between fields is needed (MI355X_MICROARCH.md, inter-workgroup visibility, form R2).  The links are
// zeroed by a memset before the launch (tag 0 = not there yet).  A wave that polls 2^22 times without progress raises the
// failure flag and publishes what it has, so every wave ends whatever happens; the flag sends the call to the fallback.
// ------------------------------------------------------------------------------------------------
#ifndef PXL_UW1_U
#define PXL_UW1_U 3            // 64-point groups per wave kept in registers
#endif
#ifndef PXL_UW1_UL
#define PXL_UW1_UL 4           // ... and kept in LDS (first in the wave's run of points)
#endif
// A link is ONE naturally aligned 8-byte word, written whole by one agent-scope relaxed store and read whole by one such load
// (data and "ready" arrive together, no fence): bits 0-1 status (0 nothing yet, 1 this chunk's AGGREGATE, 2 the inclusive PREFIX up
// to and including this chunk -- the prefix overwrites the aggregate, a reader can use either), bits 2-3 "a NaN in row 0 / 1" (of the
// chunk / up to the chunk), bits 4-33 and 34-63 the two rows' counts as 30-bit two's complement (|count| <= points so far: batches
// of 2^29 points and more take the two-pass form).  (Round 4's first form kept aggregate and the two prefixes in three words of a
// 32-byte link: three loads per link and poll.)
#ifndef PXL_UW_LINK_WORDS
#define PXL_UW_LINK_WORDS 8            // link stride in 8-byte words
#endif
struct UwLink { unsigned long long w; unsigned long long pad[PXL_UW_LINK_WORDS - 1]; };
#define PXL_UW_ONEPASS_MAX (1LL << 29)

__device__ inline unsigned long long uw_pack(int v0, int v1, unsigned int nan2, unsigned int status) {
    return (unsigned long long)(status & 3u) | ((unsigned long long)(nan2 & 3u) << 2) |
           ((unsigned long long)((unsigned int)v0 & 0x3fffffffu) << 4) | ((unsigned long long)((unsigned int)v1 & 0x3fffffffu) << 34);
}
__device__ inline void uw_publish(UwLink* p, int v0, int v1, unsigned int nan2, unsigned int status) {
    __hip_atomic_store(&p->w, uw_pack(v0, v1, nan2, status), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline unsigned long long uw_peek(const UwLink* p) {
    return __hip_atomic_load(&p->w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline int uw_field0(unsigned long long w) { return ((int)((unsigned int)(w >> 4) << 2)) >> 2; }
__device__ inline int uw_field1(unsigned long long w) { return ((int)((unsigned int)(w >> 34) << 2)) >> 2; }

// The exclusive carry of link `id` (one wave): lane l reads link top - l of a 64-link window (WIN windows per round trip); up to the
// nearest link that holds a prefix the aggregates are added, a link with nothing yet means poll the round again, a round without a
// prefix moves WIN windows further back.  Links before the first one count as a prefix of zero.  Polls are bounded: a wave that makes
// no progress 2^22 times gives up (the caller raises the failure flag and publishes what it has, so every wave ends).
template <int WIN>
__device__ inline void uw_lookback(const UwLink* links, int64_t id, int lane, int* E, unsigned int* nan_before, bool* gave_up) {
    E[0] = E[1] = 0; *nan_before = 0; *gave_up = false;
    if (id <= 0) return;
    int64_t top = id - 1;
    unsigned int polls = 0;
    for (;;) {
        unsigned long long w[WIN];
#pragma unroll
        for (int q = 0; q < WIN; ++q) {
            const int64_t j = top - 64 * q - lane;
            w[q] = j >= 0 ? uw_peek(&links[j]) : 2ull;              // status 2, counts 0
        }
        int e0 = 0, e1 = 0;
        unsigned int en = 0;
        int state = 0;                   // 0: no prefix met yet; 1: done; 2: a link before the first prefix holds nothing yet
#pragma unroll
        for (int q = 0; q < WIN; ++q) {
            const unsigned int st = (unsigned int)w[q] & 3u;
            const unsigned long long pmask = __ballot(st == 2u), amask = __ballot(st != 0u);
            const int first = pmask ? __builtin_ctzll(pmask) : 64;          // nearest link of this window with a prefix
            const unsigned long long need = first >= 64 ? ~0ull : ((1ull << first) - 1ull);
            const bool mine = lane <= first;                                 // aggregates before it and the prefix itself
            const int s0 = uw_wave_total(mine ? uw_field0(w[q]) : 0), s1 = uw_wave_total(mine ? uw_field1(w[q]) : 0);
            const unsigned int nb = mine ? ((unsigned int)w[q] >> 2) & 3u : 0u;
            const unsigned int nbw = (__ballot(nb & 1u) != 0ull ? 1u : 0u) | (__ballot(nb & 2u) != 0ull ? 2u : 0u);
            if (state == 0) {
                if ((amask & need) != need) state = 2;
                else { e0 += s0; e1 += s1; en |= nbw; if (first < 64) state = 1; }
            }
        }
        if (state == 2) {
            if (++polls > (1u << 22)) { *gave_up = true; return; }
            __builtin_amdgcn_s_sleep(2);
            continue;
        }
        E[0] += e0; E[1] += e1; *nan_before |= en;
        if (state == 1) return;
        top -= 64 * WIN;
        polls = 0;
    }
}

// PXL_UW1_WAVES waves of a workgroup share one chunk (64 * (PXL_UW1_U + PXL_UW1_UL) points each, back to back) and ONE link: their sums
// meet in LDS and only wave 0 walks the links.  (A first version gave every wave its own link: 195 000 links for 1e8 points, 5 000 of them
// resident at once, and a wave that starts with all its predecessors still computing walks back through dozens of 64-link
// windows -- 2.4 ms against the two-pass form's 1.05; profiles/r04_unwind_onepass.txt.  With 8 waves per link there are 16 times
// fewer links, ~500 resident, and a look-back is a handful of windows: 0.93 ms; 16 waves of 256 points each: 0.88 ms.  Letting ALL
// waves of the workgroup look back at once -- 1 024 links per round, windows combined through LDS -- was built and measured SLOWER
// (1.10 ms: sixteen times the polling traffic and two workgroup barriers per round) and removed.)
// Chunk size (round 4, late): the hand-over costs ~3.5 us per chunk whatever its size (0.175 of 0.855 ms at 4 096 points), so the
// chunk should be as large as two workgroups per CU allow.  Registers hold 3 groups of 64 points per wave (60 VGPRs), LDS the 4
// groups before them (17 B per point: 70 KB of the CU's 160 per workgroup): 7 168 points per link, 0.78-0.80 ms where the
// 4 096-point chunk took 0.87 on the same box.  Other splits on that box: 4 + 4 (72 VGPRs: one workgroup per CU) 0.77-0.81,
// 8 + 0 (80 VGPRs, one per CU) 0.80-0.82, 2 + 4 0.82, 3 + 3 0.80-0.82, 4 + 0 0.875 (tools/research/r04_24.sh).
#ifndef PXL_UW1_WAVES
#define PXL_UW1_WAVES 16
#endif
#ifndef PXL_UW1_CHUNK
#define PXL_UW1_CHUNK (64LL * (PXL_UW1_U + PXL_UW1_UL) * PXL_UW1_WAVES)       // points per workgroup and link
#endif
// (Chunks of 13 312 points with ONE workgroup per CU -- 4 groups per wave in registers, 9 in LDS -- were 3-4 % faster than these while a
// link was three words; with single-word links the two-workgroup form is the faster one: 0.70-0.71 against 0.71-0.74 ms.)
#ifndef PXL_UW1_WIN
#define PXL_UW1_WIN 1          // 64-link windows wave 0 reads per look-back round.  More windows per round are slower whether or not the
#endif                         // registers allow two workgroups per CU (more polling traffic): 1 / 2 / 3 windows 0.84 / 0.89-0.91 / 0.92-0.95 ms at
                               // 46 / 54 / 63 VGPRs (4 096-point chunks, round 4, one box; tools/research/r04_23.sh)
// The per-wave sums of a chunk, gathered by lanes 0..NW-1 and reduced in the wave: totals, the part before wave `upto`, and the
// NaN flags likewise.  (An unrolled scalar loop over the LDS arrays keeps 3 NW values in flight in as many registers, and a
// loop up to `wave` is a chain of LDS round trips.)
struct UwGather { int tot[2], before[2]; unsigned int nan_all, nan_before; };
template <int NW>
__device__ inline UwGather uw_gather(const int (*wsum)[NW], const int* wnan, int lane, int upto) {
    const bool in = lane < NW;
    const int a0 = in ? wsum[0][lane] : 0, a1 = in ? wsum[1][lane] : 0;
    const unsigned int na = in ? (unsigned)wnan[lane] : 0u;
    const int s0 = uw_scan64(a0), s1 = uw_scan64(a1);
    UwGather g;
    g.tot[0] = __builtin_amdgcn_readlane(s0, 63); g.tot[1] = __builtin_amdgcn_readlane(s1, 63);
    g.before[0] = upto > 0 ? __builtin_amdgcn_readlane(s0, upto - 1) : 0;
    g.before[1] = upto > 0 ? __builtin_amdgcn_readlane(s1, upto - 1) : 0;
    const unsigned long long n0 = __ballot(na & 1u), n1 = __ballot(na & 2u);
    const unsigned long long below = (1ull << upto) - 1ull;                 // upto < 64
    g.nan_all = (n0 ? 1u : 0u) | (n1 ? 2u : 0u);
    g.nan_before = ((n0 & below) ? 1u : 0u) | ((n1 & below) ? 2u : 0u);
    return g;
}

template <class SRC, int U, int UL>
__global__ __launch_bounds__(64 * PXL_UW1_WAVES) void k_unwind_onepass(SRC src, typename SRC::raw_t* out, int64_t n, UwLink* __restrict__ links,
                                                                       unsigned int* __restrict__ ticket, int32_t* __restrict__ flag) {
    constexpr int NROW = SRC::NROW, NW = PXL_UW1_WAVES;
    __shared__ unsigned int id_s;
    __shared__ int wsum_s[2][NW], wnan_s[NW];
    __shared__ int excl_s[2];
    __shared__ unsigned int nanb_s, gaveup_s;
    // the first UL groups of every wave wait in LDS between the sum and the apply step (17 B per point), the last U in registers
    __shared__ double2 mL_s[UL > 0 ? UL : 1][64 * NW];
    __shared__ unsigned char ccL_s[UL > 0 ? UL : 1][64 * NW];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x == 0) { id_s = atomicAdd(ticket, 1u); gaveup_s = 0u; }       // (taking blockIdx.x instead of a ticket: 0.845 vs 0.850 ms, not worth the dispatch-order assumption)
    __syncthreads();
    const int64_t id = (int64_t)id_s;
    const int64_t base = (id * NW + wave) * 64 * (U + UL);
    const int64_t baseR = base + 64 * UL;                                         // first point of the register groups
    const double P = src.period, rP = src.rperiod, ref = src.ref;
    // this wave's points are base + idx, idx < nvalid (32-bit tests instead of 64-bit ones per group); kpos: none of them is the batch's first
    const int64_t left = n - base;
    const int nvalid = left <= 0 ? 0 : (left >= 64 * (U + UL) ? 64 * (U + UL) : (int)left);
    const bool kpos = base > 0;
    typename SRC::raw_t vl[UL > 0 ? UL : 1], v[U];
#pragma unroll
    for (int u = 0; u < UL; ++u) {
        const int idx = u * 64 + lane;
        vl[u] = (idx < nvalid) ? src.load(base + idx) : src.zero();
    }
    // (the register groups' loads are issued one by one as the parked groups are consumed: U + UL loads in flight at once cost
    // 16 more registers than two workgroups per CU leave)
    if (UL == 0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = u * 64 + lane;
            v[u] = (idx < nvalid) ? src.load(base + idx) : src.zero();
        }
    }
    double mfirst[2];
    src.to_m((base > 0 && base - 1 < n) ? src.load(base - 1) : src.zero(), mfirst);
    double mlast[2] = {mfirst[0], mfirst[1]};
    double m[U][2];
    int cc[U];                                      // (c0 + 1) | (c1 + 1) << 16: the two rows' increments as the wave scan takes them
    int sum[2] = {0, 0};
    bool nanl[2] = {false, false};
#pragma unroll
    for (int u = 0; u < UL; ++u) {
        const int idx = u * 64 + lane;
        const bool valid = idx < nvalid;
        double mm[2], mp[2];
        int c[2] = {0, 0};
        uw_element(src, lane, (int64_t)((kpos || idx > 0) ? 1 : 0), valid, vl[u], mlast, mm, mp, c);      // uw_element only asks whether k > 0
        mL_s[u][threadIdx.x] = make_double2(mm[0], mm[1]);
        ccL_s[u][threadIdx.x] = (unsigned char)((c[0] + 1) | (NROW == 2 ? (c[1] + 1) << 2 : 0));
#pragma unroll
        for (int r = 0; r < NROW; ++r) {
            sum[r] += c[r];
            nanl[r] = nanl[r] || (valid && mm[r] != mm[r]);
            mlast[r] = uw_lane63(mm[r]);
        }
#pragma unroll
        for (int u2 = u * U / UL; u2 < (u + 1) * U / UL; ++u2) {           // this group's share of the register groups' loads
            const int idx2 = (UL + u2) * 64 + lane;
            v[u2] = (idx2 < nvalid) ? src.load(base + idx2) : src.zero();
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int idx = (UL + u) * 64 + lane;
        const bool valid = idx < nvalid;
        double mp[2];
        int c[2] = {0, 0};
        uw_element(src, lane, (int64_t)((kpos || idx > 0) ? 1 : 0), valid, v[u], mlast, m[u], mp, c);
        cc[u] = (c[0] + 1) | (NROW == 2 ? (c[1] + 1) << 16 : 0);
#pragma unroll
        for (int r = 0; r < NROW; ++r) {
            sum[r] += c[r];
            nanl[r] = nanl[r] || (valid && m[u][r] != m[u][r]);
            mlast[r] = uw_lane63(m[u][r]);
        }
    }
    const unsigned int nh_wave = (__ballot(nanl[0]) != 0ull ? 1u : 0u) | (NROW == 2 && __ballot(nanl[1]) != 0ull ? 2u : 0u);
    {
        int T0 = uw_wave_total(sum[0]), T1 = NROW == 2 ? uw_wave_total(sum[1]) : 0;
        if (lane == 0) { wsum_s[0][wave] = T0; wsum_s[1][wave] = T1; wnan_s[wave] = (int)nh_wave; }
    }
    __syncthreads();
    if (wave == 0) {
        const UwGather gt = uw_gather<NW>(wsum_s, wnan_s, lane, 0);
        const int T[2] = {gt.tot[0], gt.tot[1]};
        const unsigned int nan_here = gt.nan_all;
        // 2. the aggregate
        if (lane == 0) uw_publish(&links[id], T[0], T[1], nan_here, 1u);
        // 3. look-back
        int E[2];
        unsigned int nan_before;
        bool gave_up;
        uw_lookback<PXL_UW1_WIN>(links, id, lane, E, &nan_before, &gave_up);
        // 4. the inclusive prefix replaces the aggregate (NaN bits: in this row up to and including this chunk)
        if (lane == 0) {
            uw_publish(&links[id], E[0] + T[0], E[1] + T[1], nan_before | nan_here, 2u);
            excl_s[0] = E[0]; excl_s[1] = E[1]; nanb_s = nan_before; gaveup_s = gave_up ? 1u : 0u;
        }
    }
    __syncthreads();
    // 5. apply: k_unwind_apply's arithmetic on the values kept in LDS and in registers.  y' = m - r P of the element before (the
    // reference's y[k-1]) is taken from the neighbouring lane's own result instead of being formed a second time from its m.
    const UwGather gi = uw_gather<NW>(wsum_s, wnan_s, lane, wave);
    int carry[2] = {excl_s[0] + gi.before[0], excl_s[1] + gi.before[1]};
    const unsigned int nan_before = nanb_s | gi.nan_before;
    bool pex[2] = {(nan_before & 1u) != 0, (nan_before & 2u) != 0};
    bool bad = gaveup_s != 0u;
    double ylast[2] = {mfirst[0] - (double)carry[0] * P, mfirst[1] - (double)carry[1] * P};       // y' of the element before this wave's first
    auto apply_group = [&](auto nonan_tag, int64_t k, const double* mv, int ccv) {
        // the plain case: no NaN in this wave's points or before them (no poison bookkeeping), every point exists, none is the first
        constexpr bool NONAN = decltype(nonan_tag)::value, FULL = NONAN;
        const bool valid = FULL || k < n;
        const int s = uw_scan64(ccv);
        const int tot = __builtin_amdgcn_readlane(s, 63);
        double y[2] = {0.0, 0.0};
#pragma unroll
        for (int r = 0; r < NROW; ++r) {
            const int field = r == 0 ? (s & 0xffff) : (s >> 16);
            const int rr = carry[r] + field - (lane + 1);            // r_k (0 at k = 0: no carry, increment forced to 0)
            carry[r] += (r == 0 ? (tot & 0xffff) : (tot >> 16)) - 64;
            const double rrd = (double)rr;
            const double yq = mv[r] - rrd * P;                       // y[k] before ref is added; k = 0: m - 0 = m, bit for bit
            const double yprev = uw_shr1_first(yq, ylast[r]);        // y[k-1] as the reference forms it
            ylast[r] = uw_lane63(yq);
            bool poisoned = false;
            if (!NONAN) {
                const unsigned long long nanmask = __ballot(valid && mv[r] != mv[r]);
                poisoned = pex[r] || (nanmask & ((2ull << lane) - 1ull)) != 0ull;
                pex[r] = pex[r] || nanmask != 0ull;
            }
            if (!valid) continue;
            if (poisoned) { y[r] = __builtin_nan("") + ref; continue; }
            if (FULL || k > 0) {
                const double a = mv[r] - yprev;
                const double qa = a * rP;
                if (!(fabs(qa - rrd) < 0.4999)) {
                    const double q = a / P;
                    if (!(rint(q) == rrd)) bad = true;
                }
            }
            y[r] = yq + ref;
        }
        if (valid) SRC::store(out, k, y);
    };
    auto apply_all = [&](auto nonan_tag) {
#pragma unroll
        for (int u = 0; u < UL; ++u) {
            const double2 mm = mL_s[u][threadIdx.x];
            const int cb = (int)ccL_s[u][threadIdx.x];
            const double mv[2] = {mm.x, mm.y};
            apply_group(nonan_tag, base + (int64_t)u * 64 + lane, mv, (cb & 3) | ((cb >> 2) << 16));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) apply_group(nonan_tag, baseR + (int64_t)u * 64 + lane, m[u], cc[u]);
    };
    if ((nan_before | nh_wave) == 0u && base > 0 && base + 64 * (U + UL) <= n) apply_all(std::true_type{});
    else apply_all(std::false_type{});
    if (__any(bad) && lane == 0) atomicOr(flag, 1);
}

// ------------------------------------------------------------------------------------------------
// Small batches: the whole of unwind! in ONE launch of one 1024-thread block, no scratch, no follow-up launches
// (a catalogue of a few thousand positions is launch-bound: the multi-kernel form costs ~60 us of dispatches, the
// serial kernel 94 ns per point; one CU needs ~9 us per 4096-point round and pass, so this wins up to 8192 points).  The block sweeps rounds of 16 wave chunks of 64*PXL_UWB_U points; chunk sums,
// NaN flags and the carries between rounds live in LDS, m stays in registers between the sum and the scan.  The
// sweep runs twice: first verifying only, then -- if every element agreed with the reference recurrence -- storing;
// otherwise waves 0/1 run the exact serial recurrence on the untouched input.
// ------------------------------------------------------------------------------------------------
#define PXL_UWB_U 4
#define PXL_UWB_MAX 8192

template <class SRC>
__global__ __launch_bounds__(1024) void k_unwind_block(SRC src, typename SRC::raw_t* out, int64_t n,
                                                       const int32_t* __restrict__ gate) {
    // gate: as the fallback of the multi-kernel form this launch runs only if *gate != 0 (its check failed)
    constexpr int NROW = SRC::NROW;
    constexpr int U = PXL_UWB_U;
    if (gate && *gate == 0) return;
    __shared__ int wsum[2][16], wnan[2][16];
    __shared__ double wlast[2][16];
    __shared__ double mcarry[2];
    __shared__ int rcarry[2], pcarry[2], bad_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double P = src.period, rP = src.rperiod, ref = src.ref;
    // as a fallback the verification is already known to fail (same increments, same check): go straight to the
    // serial recurrence
    if (threadIdx.x == 0) bad_s = gate ? 1 : 0;
    for (int pass = gate ? 2 : 0; pass < 2; ++pass) {
        if (threadIdx.x == 0) { rcarry[0] = rcarry[1] = 0; pcarry[0] = pcarry[1] = 0; mcarry[0] = mcarry[1] = 0.0; }
        __syncthreads();
        if (pass == 1 && bad_s) break;
        bool bad = false;
        for (int64_t r0 = 0; r0 < n; r0 += 1024 * U) {
            const int64_t base = r0 + (int64_t)wave * 64 * U;
            // the m just before this chunk: inside a round the input has not been overwritten yet (reload it); across
            // rounds it may have been (in-place calls), so wave 0 takes it from the previous round's last chunk
            double mfirst[2] = {mcarry[0], mcarry[1]};
            if (wave > 0) src.to_m((base >= 1 && base - 1 < n) ? src.load(base - 1) : src.zero(), mfirst);
            double mlast[2] = {mfirst[0], mfirst[1]};
            typename SRC::raw_t v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t k = base + (int64_t)u * 64 + lane;
                v[u] = (k < n) ? src.load(k) : src.zero();
            }
            double m[U][2];
            int cc[U][2];
            int sum[2] = {0, 0};
            bool nanl[2] = {false, false};
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t k = base + (int64_t)u * 64 + lane;
                double mp[2];
                cc[u][0] = cc[u][1] = 0;
                uw_element(src, lane, k, k < n, v[u], mlast, m[u], mp, cc[u]);
#pragma unroll
                for (int r = 0; r < NROW; ++r) {
                    sum[r] += cc[u][r];
                    nanl[r] = nanl[r] || (k < n && m[u][r] != m[u][r]);
                    mlast[r] = uw_lane63(m[u][r]);
                }
            }
#pragma unroll
            for (int r = 0; r < NROW; ++r) {
                const int tot = uw_wave_total(sum[r]);
                const bool anynan = __ballot(nanl[r]) != 0ull;
                if (lane == 0) { wsum[r][wave] = tot; wnan[r][wave] = anynan; wlast[r][wave] = mlast[r]; }
            }
            __syncthreads();
            int carry[2];
            bool pex[2];
#pragma unroll
            for (int r = 0; r < NROW; ++r) {
                carry[r] = rcarry[r];
                pex[r] = pcarry[r] != 0;
                for (int w = 0; w < wave; ++w) { carry[r] += wsum[r][w]; pex[r] = pex[r] || wnan[r][w] != 0; }
            }
            mlast[0] = mfirst[0]; mlast[1] = mfirst[1];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t k = base + (int64_t)u * 64 + lane;
                const bool valid = k < n;
                int s = (cc[u][0] + 1) | (NROW == 2 ? (cc[u][1] + 1) << 16 : 0);
                s = uw_scan64(s);
                const int tot = __builtin_amdgcn_readlane(s, 63);
                double y[2];
#pragma unroll
                for (int r = 0; r < NROW; ++r) {
                    const double mp = uw_shr1_first(m[u][r], mlast[r]);
                    mlast[r] = uw_lane63(m[u][r]);
                    const int field = r == 0 ? (s & 0xffff) : (s >> 16);
                    const int rr = carry[r] + field - (lane + 1);
                    carry[r] += (r == 0 ? (tot & 0xffff) : (tot >> 16)) - 64;
                    const unsigned long long nanmask = __ballot(valid && m[u][r] != m[u][r]);
                    const bool poisoned = pex[r] || (nanmask & ((2ull << lane) - 1ull)) != 0ull;
                    pex[r] = pex[r] || nanmask != 0ull;
                    if (!valid) continue;
                    if (poisoned) { y[r] = __builtin_nan("") + ref; continue; }
                    if (k > 0) {
                        const double yprev = mp - (double)(rr - cc[u][r]) * P;
                        const double a = m[u][r] - yprev;
                        const double qa = a * rP;
                        if (!(fabs(qa - (double)rr) < 0.4999)) {
                            const double q = a / P;
                            if (!(rint(q) == (double)rr)) bad = true;
                        }
                    }
                    y[r] = (m[u][r] - (double)(k > 0 ? rr : 0) * P) + ref;
                }
                if (pass == 1 && valid) SRC::store(out, k, y);
            }
            __syncthreads();
            if (threadIdx.x == 0) {
#pragma unroll
                for (int r = 0; r < NROW; ++r) {
                    int t = rcarry[r], pz = pcarry[r];
                    for (int w = 0; w < 16; ++w) { t += wsum[r][w]; pz |= wnan[r][w]; }
                    rcarry[r] = t; pcarry[r] = pz; mcarry[r] = wlast[r][15];
                }
            }
            __syncthreads();
        }
        if (__any(bad) && lane == 0) atomicOr(&bad_s, 1);
        __syncthreads();
    }
    __syncthreads();
    if (!bad_s || wave >= NROW) return;
    // exact serial recurrence (the k_unwind_rows loop) on the untouched input: wave `row` owns coordinate row `row`
    const int row = wave;
    double prev = 0.0;
    bool have_prev = false;
    for (int64_t b0 = 0; b0 < n; b0 += 64) {
        const int64_t k = b0 + lane;
        double mm[2] = {0.0, 0.0};
        if (k < n) src.to_m(src.load(k), mm);
        const double mine = mm[row];
        double y = mine;
        const int cnt = (int)((n - b0) < 64 ? (n - b0) : 64);
        for (int l = 0; l < cnt; ++l) {
            const double ml = __shfl(mine, l, 64);
            const double yl = have_prev ? ml - rint((ml - prev) / P) * P : ml;
            prev = yl;
            have_prev = true;
            if (lane == l) y = yl;
        }
        if (k < n) reinterpret_cast<double*>(out)[NROW * k + row] = y + ref;
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
__global__ __launch_bounds__(256) void k_rewind(int64_t n, double* a, double period, double ref, int sub_ref,
                                                const int32_t* __restrict__ gate) {
    if (gate && *gate == 0) return;
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
