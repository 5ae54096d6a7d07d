// exp_scalar_gather.cpp -- does the SCALAR memory path (s_load through the scalar data cache into the L2) offer random-access
// throughput on top of what the vector path (TA / TCP) delivers?  The scattered sampler (config 5) is bound by the rate of
// per-lane divergent accesses; scalar loads do not go through the TCP.  Measures random 8-byte loads per second from buffers
// of several sizes: per lane (vector), per wave (scalar, 8 or 16 in flight) and both kinds at once.
//   hipcc --offload-arch=gfx950 -O2 tools/research/exp_scalar_gather.cpp -o tools/research/exp_scalar_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__device__ inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// vector: every lane issues `iters` rounds of 8 independent random 8-byte loads
__global__ __launch_bounds__(256) void k_vec(const uint64_t* __restrict__ buf, uint32_t mask, int iters, uint64_t* out) {
    uint32_t s = mix(blockIdx.x * 256u + threadIdx.x + 1u);
    uint64_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint64_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { s = s * 1664525u + 1013904223u; v[u] = __builtin_nontemporal_load(buf + (mix(s) & mask)); }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u];
    }
    if (acc == 0x1234567) out[0] = acc;
}
// scalar: every WAVE issues `iters` rounds of 8 independent random 8-byte s_loads (byte offset < 4 GiB)
__global__ __launch_bounds__(256) void k_scal(const uint64_t* __restrict__ buf, uint32_t mask, int iters, uint64_t* out) {
    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
    uint32_t s = mix(wave + 1u);
    uint64_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint32_t o[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { s = s * 1664525u + 1013904223u; o[u] = (mix(s) & mask) << 3; }
        uint64_t v0, v1, v2, v3, v4, v5, v6, v7;
        asm volatile(
            "s_load_dwordx2 %0, %8, %9\n s_load_dwordx2 %1, %8, %10\n s_load_dwordx2 %2, %8, %11\n s_load_dwordx2 %3, %8, %12\n"
            "s_load_dwordx2 %4, %8, %13\n s_load_dwordx2 %5, %8, %14\n s_load_dwordx2 %6, %8, %15\n s_load_dwordx2 %7, %8, %16\n"
            "s_waitcnt lgkmcnt(0)"
            : "=&s"(v0), "=&s"(v1), "=&s"(v2), "=&s"(v3), "=&s"(v4), "=&s"(v5), "=&s"(v6), "=&s"(v7)
            : "s"(buf), "s"(o[0]), "s"(o[1]), "s"(o[2]), "s"(o[3]), "s"(o[4]), "s"(o[5]), "s"(o[6]), "s"(o[7]) : "memory");
        acc ^= v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7;
    }
    if (acc == 0x1234567 && threadIdx.x == 0) out[0] = acc;
}
// both at once in every wave: 8 vector loads per lane and `SPER` rounds of 8 scalar loads per wave per iteration
template <int SPER>
__global__ __launch_bounds__(256) void k_both(const uint64_t* __restrict__ buf, uint32_t mask, int iters, uint64_t* out) {
    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
    uint32_t ss = mix(wave + 1u);
    uint32_t s = mix(blockIdx.x * 256u + threadIdx.x + 77u);
    uint64_t acc = 0, sacc = 0;
    for (int it = 0; it < iters; ++it) {
        uint64_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { s = s * 1664525u + 1013904223u; v[u] = __builtin_nontemporal_load(buf + (mix(s) & mask)); }
#pragma unroll
        for (int r = 0; r < SPER; ++r) {
            uint32_t o[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { ss = ss * 1664525u + 1013904223u; o[u] = (mix(ss) & mask) << 3; }
            uint64_t v0, v1, v2, v3, v4, v5, v6, v7;
            asm volatile(
                "s_load_dwordx2 %0, %8, %9\n s_load_dwordx2 %1, %8, %10\n s_load_dwordx2 %2, %8, %11\n s_load_dwordx2 %3, %8, %12\n"
                "s_load_dwordx2 %4, %8, %13\n s_load_dwordx2 %5, %8, %14\n s_load_dwordx2 %6, %8, %15\n s_load_dwordx2 %7, %8, %16\n"
                "s_waitcnt lgkmcnt(0)"
                : "=&s"(v0), "=&s"(v1), "=&s"(v2), "=&s"(v3), "=&s"(v4), "=&s"(v5), "=&s"(v6), "=&s"(v7)
                : "s"(buf), "s"(o[0]), "s"(o[1]), "s"(o[2]), "s"(o[3]), "s"(o[4]), "s"(o[5]), "s"(o[6]), "s"(o[7]) : "memory");
            sacc ^= v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u];
    }
    if ((acc ^ sacc) == 0x1234567) out[0] = acc;
}

template <class F> static double time_ms(F launch) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    launch(); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a)); launch(); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b)); return ms;
}
int main() {
    const size_t maxbytes = (size_t)4 << 30;        // s_load offsets are 32-bit
    uint64_t *buf, *out;
    CHECK(hipMalloc(&buf, maxbytes)); CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(buf, 1, maxbytes));
    const int blocks = 256 * 8;                      // 8 blocks of 4 waves per CU
    for (double mb : {4096.0}) {          // how the two paths add up when the scalar share grows
        const uint32_t mask = (uint32_t)(mb * 1048576.0 / 8.0) - 1u;
        const int iv = 32;
        const double nv = (double)blocks * 256 * iv * 8, nb1 = (double)blocks * 4 * iv * 8;
        double t8 = time_ms([&] { hipLaunchKernelGGL((k_both<8>), dim3(blocks), dim3(256), 0, 0, buf, mask, iv, out); });
        double t16 = time_ms([&] { hipLaunchKernelGGL((k_both<16>), dim3(blocks), dim3(256), 0, 0, buf, mask, iv, out); });
        double t32 = time_ms([&] { hipLaunchKernelGGL((k_both<32>), dim3(blocks), dim3(256), 0, 0, buf, mask, iv, out); });
        double t64 = time_ms([&] { hipLaunchKernelGGL((k_both<64>), dim3(blocks), dim3(256), 0, 0, buf, mask, iv, out); });
        printf("{\"MiB\": %.1f, \"scalar_rounds_per_vector_round\": [8, 16, 32, 64], \"ms\": [%.3f, %.3f, %.3f, %.3f], \"vector_G\": [%.2f, %.2f, %.2f, %.2f], "
               "\"scalar_G\": [%.2f, %.2f, %.2f, %.2f]}\n", mb, t8, t16, t32, t64, nv / t8 / 1e6, nv / t16 / 1e6, nv / t32 / 1e6, nv / t64 / 1e6,
               8 * nb1 / t8 / 1e6, 16 * nb1 / t16 / 1e6, 32 * nb1 / t32 / 1e6, 64 * nb1 / t64 / 1e6);
        fflush(stdout);
    }
    for (double mb : {0.5, 2.0, 64.0, 1024.0, 4096.0}) {
        const uint32_t mask = (uint32_t)(mb * 1048576.0 / 8.0) - 1u;
        const int iv = 64, is = 512;
        double tv = time_ms([&] { hipLaunchKernelGGL(k_vec, dim3(blocks), dim3(256), 0, 0, buf, mask, iv, out); });
        double ts = time_ms([&] { hipLaunchKernelGGL(k_scal, dim3(blocks), dim3(256), 0, 0, buf, mask, is, out); });
        double tb1 = time_ms([&] { hipLaunchKernelGGL((k_both<1>), dim3(blocks), dim3(256), 0, 0, buf, mask, iv, out); });
        double tb4 = time_ms([&] { hipLaunchKernelGGL((k_both<4>), dim3(blocks), dim3(256), 0, 0, buf, mask, iv, out); });
        const double nv = (double)blocks * 256 * iv * 8, ns = (double)blocks * 4 * is * 8;
        const double nb1 = (double)blocks * 4 * iv * 8, nb4 = nb1 * 4;
        printf("{\"MiB\": %.1f, \"vector_Gloads_s\": %.2f, \"scalar_Gloads_s\": %.3f, \"both1\": {\"ms\": %.3f, \"vector_G\": %.2f, \"scalar_G\": %.3f}, "
               "\"both4\": {\"ms\": %.3f, \"vector_G\": %.2f, \"scalar_G\": %.3f}, \"vec_ms\": %.3f, \"scal_ms\": %.3f}\n",
               mb, nv / tv / 1e6, ns / ts / 1e6, tb1, nv / tb1 / 1e6, nb1 / tb1 / 1e6, tb4, nv / tb4 / 1e6, nb4 / tb4 / 1e6, tv, ts);
        fflush(stdout);
    }
    return 0;
}
