// native_bench.cpp -- the hot path driven through the C ABI alone (no Python, no torch): allocates with
// hipMalloc, fills a synthetic full-sky CAR map, reprojects it, checks a partition-of-unity property and
// prints the HIP-event kernel time and achieved algorithmic GB/s.  Useful under rocprofv3 without an
// interpreter in the way, and as the smallest example of a host that is not the Python mirror.
//
//   hipcc --offload-arch=gfx950 -O2 -I include tools/native/native_bench.cpp -L pixell.jl_amd -lpixell_hip \
//         -Wl,-rpath,$PWD/pixell.jl_amd -o native_bench && ./native_bench 43200 3 same 20 [placed [headroom GiB]]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "pixell_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_PXL(x) do { int rc_ = (x); if (rc_ != 0) { char m_[256]; pxl_last_error(m_, sizeof m_); fprintf(stderr, "%s -> %d: %s\n", #x, rc_, m_); return 3; } } while (0)

// fullsky_geometry(res) of the reference (enmap_geom.jl:47-67) for nx pixels around the sky
static void fullsky(int64_t nx, int64_t shape[2], pxl_car_wcs* w) {
    const double pi = 3.141592653589793;
    double res = 2 * pi / (double)nx;
    shape[0] = (int64_t)nearbyint(2 * pi / res);
    shape[1] = (int64_t)nearbyint(pi / res + 1);
    w->cdelt[0] = -360.0 / (double)shape[0];
    w->cdelt[1] = 180.0 / (double)(shape[1] - 1);
    w->crpix[0] = floor((double)shape[0] / 2) + 0.5;
    w->crpix[1] = (double)(shape[1] + 1) / 2;
    w->crval[0] = res * 90 / pi;
    w->crval[1] = 0.0;
    w->unit = pi / 180;
}

int main(int argc, char** argv) {
    int64_t nx = argc > 1 ? atoll(argv[1]) : 4096;
    int64_t nc = argc > 2 ? atoll(argv[2]) : 1;
    bool refine = argc > 3 && strcmp(argv[3], "refine") == 0;     // "same": half-pixel shift; "refine": 2x grid
    int steps = argc > 4 ? atoi(argv[4]) : 10;
    // "placed GiB": the maps from pxl_mem_pair_alloc with that much head-room (class-aware placement through the C ABI)
    const bool placed = argc > 5 && strcmp(argv[5], "placed") == 0;
    const uint64_t headroom = (uint64_t)(argc > 6 ? atoll(argv[6]) : 144) << 30;

    int64_t shape_in[3], shape_out[2];
    pxl_car_wcs win, wout;
    fullsky(nx, shape_in, &win);
    shape_in[2] = nc;
    if (refine) fullsky(2 * nx, shape_out, &wout);
    else { shape_out[0] = shape_in[0]; shape_out[1] = shape_in[1]; wout = win; wout.crpix[0] += 0.5; wout.crpix[1] += 0.5; }

    const size_t n_src = (size_t)shape_in[0] * shape_in[1] * nc, n_dst = (size_t)shape_out[0] * shape_out[1] * nc;
    double *src = nullptr, *dst = nullptr;
    hipStream_t st;
    CHECK_HIP(hipStreamCreate(&st));
    pxl_mem_pair pair;
    memset(&pair, 0, sizeof pair);
    if (placed) {
        CHECK_PXL(pxl_mem_pair_alloc(n_src * 8, n_dst * 8, headroom, &pair, st));
        src = (double*)pair.src; dst = (double*)pair.dst;
    } else {
        CHECK_HIP(hipMalloc(&src, n_src * 8));
        CHECK_HIP(hipMalloc(&dst, n_dst * 8));
    }

    pxl_reproject_plan* plan = nullptr;
    CHECK_PXL(pxl_reproject_plan_create(&win, shape_in, 0, shape_in[1], &wout, shape_out, 0, shape_out[1], &plan));

    // property check first: a constant map reprojects to the same constant (weights sum to one on a full-sky map)
    std::vector<double> ones(shape_in[0], 1.0);
    for (int64_t r = 0; r < shape_in[1] * nc; ++r)
        CHECK_HIP(hipMemcpyAsync(src + r * shape_in[0], ones.data(), shape_in[0] * 8, hipMemcpyHostToDevice, st));
    CHECK_PXL(pxl_reproject_execute(plan, src, dst, st));
    std::vector<double> row(shape_out[0]);
    CHECK_HIP(hipMemcpyAsync(row.data(), dst + (size_t)(shape_out[1] / 2) * shape_out[0], shape_out[0] * 8, hipMemcpyDeviceToHost, st));
    CHECK_HIP(hipStreamSynchronize(st));
    double worst = 0;
    for (double v : row) worst = fmax(worst, fabs(v - 1.0));
    if (!(worst < 1e-13)) { fprintf(stderr, "partition of unity violated: %g\n", worst); return 4; }

    for (int64_t c = 0; c < nc; ++c)
        CHECK_PXL(pxl_fill_random_f64(src + (size_t)c * shape_in[0] * shape_in[1], shape_in[0] * shape_in[1], 1234 + c, 0, 0, st));
    CHECK_PXL(pxl_reproject_build_tables(plan, st));
    for (int k = 0; k < 3; ++k) CHECK_PXL(pxl_reproject_execute_rows(plan, src, dst, 0, shape_out[1], st));
    hipEvent_t e0, e1;
    CHECK_HIP(hipEventCreate(&e0));
    CHECK_HIP(hipEventCreate(&e1));
    CHECK_HIP(hipEventRecord(e0, st));
    for (int k = 0; k < steps; ++k) CHECK_PXL(pxl_reproject_execute_rows(plan, src, dst, 0, shape_out[1], st));
    CHECK_HIP(hipEventRecord(e1, st));
    CHECK_HIP(hipEventSynchronize(e1));
    float ms = 0;
    CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
    ms /= (float)steps;
    double bytes = 8.0 * (double)(n_src + n_dst);
    printf("{\"native\": true, \"shape_in\": [%lld, %lld, %lld], \"shape_out\": [%lld, %lld], \"kernel_ms\": %.4f, "
           "\"algorithmic_GBps\": %.1f, \"frac_of_8TBps\": %.4f, \"Mpix_per_s\": %.1f, \"unity_err\": %.3g, \"placed\": %s, "
           "\"classes\": %d, \"dst_two_classes\": %d, \"src_own_class\": %d, \"allocation_GiB\": %.1f}\n",
           (long long)shape_in[0], (long long)shape_in[1], (long long)nc, (long long)shape_out[0], (long long)shape_out[1],
           ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000.0, (double)n_dst / ms / 1e3, worst, placed ? "true" : "false",
           pair.classes, pair.dst_two_classes, pair.src_own_class, pair.arena_bytes / 1073741824.0);
    pxl_reproject_plan_destroy(plan);
    if (placed) CHECK_PXL(pxl_mem_pair_free(&pair));
    else { (void)hipFree(src); (void)hipFree(dst); }
    return 0;
}
