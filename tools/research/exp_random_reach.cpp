// exp_random_reach.cpp -- random-access rate of the vector path against the FOOTPRINT (0.5 MiB ... 48 GiB): where do the L2,
// the Infinity Cache and the address translation stop helping?  Each lane issues rounds of 8 independent random accesses;
// an access is one 8-byte load, or a 32-byte "cell" (two adjacent 16-byte loads, 16-byte aligned: what a 2x2 bilinear
// neighbourhood costs in the row-pair layout).  Prints G accesses/s.
//   hipcc --offload-arch=gfx950 -O2 tools/research/exp_random_reach.cpp -o tools/research/exp_random_reach
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
__device__ inline uint64_t mix64(uint64_t z) { z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; return z ^ (z >> 31); }

template <int MODE>      // 0: 8-byte load; 1: 32-byte cell
__global__ __launch_bounds__(256) void k_rand(const double2* __restrict__ buf, uint64_t ncell16, int iters, double* out) {
    uint64_t s = mix64((uint64_t)blockIdx.x * 256u + threadIdx.x + 1u);
    double acc = 0.0;
    for (int it = 0; it < iters; ++it) {
        double2 a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s = s * 6364136223846793005ULL + 1442695040888963407ULL;
            // multiply-shift range reduction (no modulo): uniform cell index below ncell16 - 1
            const uint64_t c = (uint64_t)(((unsigned __int128)mix64(s) * (ncell16 - 1)) >> 64);
            if (MODE == 0) { a[u].x = __builtin_nontemporal_load(reinterpret_cast<const double*>(buf + c)); a[u].y = 0.0; b[u] = a[u]; }
            else { a[u] = buf[c]; b[u] = buf[c + 1]; }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += a[u].x + a[u].y + b[u].x + b[u].y;
    }
    if (acc == 1.2345) out[0] = acc;
}
// the sampler's shape: the cell index arrives in a coalesced 16-byte-per-point stream, SUNR points per lane per trip,
// one 8-byte result per point written coalesced; PREFETCH loads the next trip's stream entries before this trip's cells
struct Pt { uint64_t cell; uint64_t pad; };
__global__ __launch_bounds__(256) void k_fill_pts(Pt* pts, int64_t n, uint64_t ncell16) {
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256)
        pts[k].cell = (uint64_t)(((unsigned __int128)mix64(k + 12345) * (ncell16 - 1)) >> 64), pts[k].pad = k;
}
__global__ __launch_bounds__(256) void k_even_cells(Pt* pts, int64_t n) {
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) pts[k].cell &= ~1ULL;
}
template <int SUNR, bool PREFETCH>
__global__ __launch_bounds__(256) void k_stream_cells(const double2* __restrict__ buf, const Pt* __restrict__ pts, int64_t n,
                                                      double* __restrict__ out) {
    const int64_t chunk = 256 * SUNR, stride = (int64_t)gridDim.x * chunk;
    int64_t k0 = (int64_t)blockIdx.x * chunk + threadIdx.x;
    Pt nxt[SUNR];
    if (PREFETCH) {
#pragma unroll
        for (int u = 0; u < SUNR; ++u) { const int64_t k = k0 + u * 256; nxt[u] = k < n ? pts[k] : Pt{0, 0}; }
    }
    for (; k0 < n; k0 += stride) {
        Pt cur[SUNR];
#pragma unroll
        for (int u = 0; u < SUNR; ++u) {
            if (PREFETCH) cur[u] = nxt[u];
            else { const int64_t k = k0 + u * 256; cur[u] = k < n ? pts[k] : Pt{0, 0}; }
        }
        double2 a[SUNR], b[SUNR];
#pragma unroll
        for (int u = 0; u < SUNR; ++u) { a[u] = buf[cur[u].cell]; b[u] = buf[cur[u].cell + 1]; }
        if (PREFETCH) {
#pragma unroll
            for (int u = 0; u < SUNR; ++u) { const int64_t k = k0 + stride + u * 256; nxt[u] = k < n ? pts[k] : Pt{0, 0}; }
        }
#pragma unroll
        for (int u = 0; u < SUNR; ++u) { const int64_t k = k0 + u * 256; if (k < n) out[k] = a[u].x + a[u].y + b[u].x + b[u].y; }
    }
}
template <class F> static double time_ms(F launch) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    launch(); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a)); launch(); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b)); return ms;
}
int main(int argc, char** argv) {
    const double max_gib = argc > 1 ? atof(argv[1]) : 48.0;
    const size_t maxbytes = (size_t)(max_gib * 1073741824.0);
    double2* buf; double* out;
    CHECK(hipMalloc(&buf, maxbytes)); CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(buf, 0, maxbytes));
    if (argc > 2) {              // the sampler-shaped variants on a 15 GiB footprint (the row-pair layout of the 0.5' map)
        const int64_t n = 400000000;
        const uint64_t ncell = (uint64_t)(15360.0 * 1048576.0 / 16.0);
        Pt* pts; double* res;
        CHECK(hipMalloc(&pts, n * sizeof(Pt))); CHECK(hipMalloc(&res, n * 8));
        hipLaunchKernelGGL(k_fill_pts, dim3(4096), dim3(256), 0, 0, pts, n, ncell);
        CHECK(hipDeviceSynchronize());
        {   // 32-byte ALIGNED cells (one sector each) on a 30 GiB footprint: the "quad" layout (all four taps of a cell together)
            const uint64_t nq = (uint64_t)(30720.0 * 1048576.0 / 16.0);
            hipLaunchKernelGGL(k_fill_pts, dim3(4096), dim3(256), 0, 0, pts, n, nq);
            hipLaunchKernelGGL(k_even_cells, dim3(4096), dim3(256), 0, 0, pts, n);
            CHECK(hipDeviceSynchronize());
            double q1 = time_ms([&] { hipLaunchKernelGGL((k_stream_cells<1, false>), dim3(4096), dim3(256), 0, 0, buf, pts, n, res); });
            double q4 = time_ms([&] { hipLaunchKernelGGL((k_stream_cells<4, false>), dim3(4096), dim3(256), 0, 0, buf, pts, n, res); });
            double q8 = time_ms([&] { hipLaunchKernelGGL((k_stream_cells<8, true>), dim3(4096), dim3(256), 0, 0, buf, pts, n, res); });
            printf("{\"aligned_cells_30GiB_G_per_s\": {\"sunr1\": %.2f, \"sunr4\": %.2f, \"sunr8_prefetch\": %.2f}}\n", n / q1 / 1e6, n / q4 / 1e6, n / q8 / 1e6);
            hipLaunchKernelGGL(k_fill_pts, dim3(4096), dim3(256), 0, 0, pts, n, ncell);
            CHECK(hipDeviceSynchronize());
        }
        for (int g : {4096}) {
            double t4 = time_ms([&] { hipLaunchKernelGGL((k_stream_cells<4, false>), dim3(g), dim3(256), 0, 0, buf, pts, n, res); });
            double t8 = time_ms([&] { hipLaunchKernelGGL((k_stream_cells<8, false>), dim3(g), dim3(256), 0, 0, buf, pts, n, res); });
            double p4 = time_ms([&] { hipLaunchKernelGGL((k_stream_cells<4, true>), dim3(g), dim3(256), 0, 0, buf, pts, n, res); });
            double p8 = time_ms([&] { hipLaunchKernelGGL((k_stream_cells<8, true>), dim3(g), dim3(256), 0, 0, buf, pts, n, res); });
            double t2 = time_ms([&] { hipLaunchKernelGGL((k_stream_cells<2, false>), dim3(g), dim3(256), 0, 0, buf, pts, n, res); });
            double t1 = time_ms([&] { hipLaunchKernelGGL((k_stream_cells<1, false>), dim3(g), dim3(256), 0, 0, buf, pts, n, res); });
            printf("{\"grid\": %d, \"stream_cells_G_per_s\": {\"sunr1\": %.2f, \"sunr2\": %.2f, \"sunr4\": %.2f, \"sunr8\": %.2f, \"sunr4_prefetch\": %.2f, \"sunr8_prefetch\": %.2f}}\n",
                   g, n / t1 / 1e6, n / t2 / 1e6, n / t4 / 1e6, n / t8 / 1e6, n / p4 / 1e6, n / p8 / 1e6);
            fflush(stdout);
        }
        return 0;
    }
    const int blocks = 256 * 8, iters = 32;
    const double n = (double)blocks * 256 * iters * 8;
    for (double mib : {0.5, 2.0, 8.0, 32.0, 128.0, 512.0, 2048.0, 4096.0, 7680.0, 12288.0, 15360.0, 24576.0, 32768.0, 49152.0}) {
        if (mib * 1048576.0 > (double)maxbytes) break;
        const uint64_t ncell = (uint64_t)(mib * 1048576.0 / 16.0);
        double t0 = time_ms([&] { hipLaunchKernelGGL((k_rand<0>), dim3(blocks), dim3(256), 0, 0, buf, ncell, iters, out); });
        double t1 = time_ms([&] { hipLaunchKernelGGL((k_rand<1>), dim3(blocks), dim3(256), 0, 0, buf, ncell, iters, out); });
        printf("{\"MiB\": %.1f, \"load8_G_per_s\": %.2f, \"cell32_G_per_s\": %.2f}\n", mib, n / t0 / 1e6, n / t1 / 1e6);
        fflush(stdout);
    }
    return 0;
}
