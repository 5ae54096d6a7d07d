// pxl_misc.h -- FITS byte-order staging and the synthetic-data generators; included by pxl_kernels.hip (one translation unit, -ffp-contract=off).
#pragma once

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
// 4-byte elements: the swap is its own inverse (BITPIX -32 <-> native Float32, either direction)
__global__ __launch_bounds__(256) void k_bswap32(const uint32_t* raw, uint32_t* dst, int64_t n) {
    const int64_t chunk = (int64_t)blockDim.x * 4;
    for (int64_t k0 = (int64_t)blockIdx.x * chunk + threadIdx.x; k0 < n; k0 += (int64_t)gridDim.x * chunk) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int64_t k = k0 + u * blockDim.x;
            if (k < n) dst[k] = __builtin_bswap32(raw[k]);
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
