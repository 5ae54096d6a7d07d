// exp_placement_vmm.cpp -- does WHERE the 22 GB maps land physically move the headline kernel, and can a job steer it?
// The cfg4 reprojection (43200 x 21601 x 3, same-resolution shifted grid) is timed on buffers obtained five ways:
//   malloc      two hipMalloc calls (what torch does), repeated with a random ballast allocation in between
//   arena       ONE hipMalloc of src + dst, sub-allocated (dst right behind src, 2 MiB aligned)
//   vmm         hipMemAddressReserve + one hipMemCreate / hipMemMap per 2 MiB (minimum granularity), per 64 MiB, per 512 MiB
//               (hipMemSetAccess refused 1 GiB and larger handles on this ROCm build)
// Prints one JSON line per trial.  No Python, no torch: the C ABI alone.
//   hipcc --offload-arch=gfx950 -O2 -I include tools/research/exp_placement_vmm.cpp -L pixell.jl_amd -lpixell_hip \
//         -Wl,-rpath,$PWD/pixell.jl_amd -o exp_placement_vmm && ./exp_placement_vmm 4
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "pixell_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
#define CHECK_PXL(x) do { int rc_ = (x); if (rc_ != 0) { char m_[256]; pxl_last_error(m_, sizeof m_); fprintf(stderr, "%s -> %d: %s\n", #x, rc_, m_); exit(3); } } while (0)

static const int64_t NX = 43200, NY = 21601, NC = 3;
static pxl_reproject_plan* g_plan;
static pxl_reproject_plan *g_plan_full, *g_plan_loads, *g_plan_stores;
static const int kChunks[] = {169, 676, 2704, 10816, 43264};      // tiles per XCD piece (169 tiles = one 16-row band of one plane)
static pxl_reproject_plan* g_plan_chunk[5];
static pxl_reproject_plan *g_plan_st_rr, *g_plan_st_chunk[5];
static pxl_reproject_plan *g_plan_ord[2], *g_plan_st_ord[2];        // tile orders 1 (odd XCDs downwards) and 2 (staggered starts): full / stores only      // stores only: round-robin tiles (one front) / XCD pieces
static hipStream_t g_st;

static double time_kernel(double* src, double* dst) {
    for (int64_t c = 0; c < NC; ++c) CHECK_PXL(pxl_fill_random_f64(src + (size_t)c * NX * NY, NX * NY, 1234 + c, 0, 0, g_st));
    CHECK_PXL(pxl_reproject_build_tables(g_plan, g_st));
    for (int k = 0; k < 3; ++k) CHECK_PXL(pxl_reproject_execute_rows(g_plan, src, dst, 0, NY, g_st));
    hipEvent_t e0, e1;
    CHECK_HIP(hipEventCreate(&e0)); CHECK_HIP(hipEventCreate(&e1));
    CHECK_HIP(hipEventRecord(e0, g_st));
    for (int k = 0; k < 10; ++k) CHECK_PXL(pxl_reproject_execute_rows(g_plan, src, dst, 0, NY, g_st));
    CHECK_HIP(hipEventRecord(e1, g_st));
    CHECK_HIP(hipEventSynchronize(e1));
    float ms = 0;
    CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
    CHECK_HIP(hipEventDestroy(e0)); CHECK_HIP(hipEventDestroy(e1));
    return ms / 10.0;
}

struct VmmBuf { void* va = nullptr; size_t bytes = 0; std::vector<hipMemGenericAllocationHandle_t> handles; size_t chunk = 0; };
static VmmBuf vmm_alloc(size_t bytes, size_t chunk) {
    VmmBuf b;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CHECK_HIP(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    if (chunk == 0) chunk = (bytes + gran - 1) / gran * gran;            // one handle
    chunk = (chunk + gran - 1) / gran * gran;
    b.bytes = (bytes + chunk - 1) / chunk * chunk;
    b.chunk = chunk;
    CHECK_HIP(hipMemAddressReserve(&b.va, b.bytes, 0, nullptr, 0));
    for (size_t off = 0; off < b.bytes; off += chunk) {
        hipMemGenericAllocationHandle_t h;
        CHECK_HIP(hipMemCreate(&h, chunk, &prop, 0));
        CHECK_HIP(hipMemMap((char*)b.va + off, chunk, 0, h, 0));
        b.handles.push_back(h);
    }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CHECK_HIP(hipMemSetAccess(b.va, b.bytes, &acc, 1));
    return b;
}
static void vmm_free(VmmBuf& b) {
    CHECK_HIP(hipMemUnmap(b.va, b.bytes));
    for (auto h : b.handles) CHECK_HIP(hipMemRelease(h));
    CHECK_HIP(hipMemAddressFree(b.va, b.bytes));
}

int main(int argc, char** argv) {
    const int trials = argc > 1 ? atoi(argv[1]) : 3;
    const double pi = 3.141592653589793;
    pxl_car_wcs win = {{-360.0 / NX, 180.0 / (NY - 1)}, {floor(NX / 2.0) + 0.5, (NY + 1) / 2.0}, {(2 * pi / NX) * 90 / pi, 0.0}, pi / 180};
    pxl_car_wcs wout = win;
    wout.crpix[0] += 0.5; wout.crpix[1] += 0.5;
    const int64_t shape_in[3] = {NX, NY, NC}, shape_out[2] = {NX, NY};
    CHECK_HIP(hipStreamCreate(&g_st));
    CHECK_PXL(pxl_reproject_plan_create(&win, shape_in, 0, NY, &wout, shape_out, 0, NY, &g_plan_full));
    setenv("PXL_REPROJECT_FLAGS", "2", 1);       // diagnostics of the kernel: issue the source loads only
    CHECK_PXL(pxl_reproject_plan_create(&win, shape_in, 0, NY, &wout, shape_out, 0, NY, &g_plan_loads));
    setenv("PXL_REPROJECT_FLAGS", "64", 1);      // ... the stores only
    CHECK_PXL(pxl_reproject_plan_create(&win, shape_in, 0, NY, &wout, shape_out, 0, NY, &g_plan_stores));
    unsetenv("PXL_REPROJECT_FLAGS");
    for (int k = 0; k < 5; ++k) {
        char v[32]; snprintf(v, sizeof v, "%d", kChunks[k]);
        setenv("PXL_REPROJECT_XCHUNK", v, 1);
        CHECK_PXL(pxl_reproject_plan_create(&win, shape_in, 0, NY, &wout, shape_out, 0, NY, &g_plan_chunk[k]));
    }
    unsetenv("PXL_REPROJECT_XCHUNK");
    setenv("PXL_REPROJECT_FLAGS", "68", 1);      // stores only + no XCD remap (tile = block index: one write front, all XCDs interleaved)
    CHECK_PXL(pxl_reproject_plan_create(&win, shape_in, 0, NY, &wout, shape_out, 0, NY, &g_plan_st_rr));
    setenv("PXL_REPROJECT_FLAGS", "64", 1);
    for (int k = 0; k < 5; ++k) {
        char v[32]; snprintf(v, sizeof v, "%d", kChunks[k]);
        setenv("PXL_REPROJECT_XCHUNK", v, 1);
        CHECK_PXL(pxl_reproject_plan_create(&win, shape_in, 0, NY, &wout, shape_out, 0, NY, &g_plan_st_chunk[k]));
    }
    unsetenv("PXL_REPROJECT_XCHUNK");
    const char* ordflags[2][2] = {{"8", "72"}, {"16", "80"}};
    for (int o = 0; o < 2; ++o) {
        setenv("PXL_REPROJECT_FLAGS", ordflags[o][0], 1);
        CHECK_PXL(pxl_reproject_plan_create(&win, shape_in, 0, NY, &wout, shape_out, 0, NY, &g_plan_ord[o]));
        setenv("PXL_REPROJECT_FLAGS", ordflags[o][1], 1);
        CHECK_PXL(pxl_reproject_plan_create(&win, shape_in, 0, NY, &wout, shape_out, 0, NY, &g_plan_st_ord[o]));
    }
    unsetenv("PXL_REPROJECT_FLAGS");
    g_plan = g_plan_full;
    const size_t bytes = (size_t)NX * NY * NC * 8;
    if (argc > 2 && strcmp(argv[2], "pmc") == 0) {
        // for rocprofv3 --pmc: one arena; stores only into the high half, then into the low half (10 timed launches each
        // after fill + 3 warm-ups: the dispatches of k_reproject_dma come in that order); the times are printed so that
        // the counter rows can be told apart
        const size_t pad = ((bytes + (2u << 20) - 1) >> 21) << 21;
        char* a;
        CHECK_HIP(hipMalloc(&a, 2 * pad));
        double* lo = (double*)a; double* hi = (double*)(a + pad);
        g_plan = g_plan_stores;
        for (int rep = 0; rep < trials; ++rep) {
            printf("{\"how\": \"pmc run, STORES ONLY to the high half\", \"rep\": %d, \"kernel_ms\": %.4f}\n", rep, time_kernel(lo, hi));
            printf("{\"how\": \"pmc run, STORES ONLY to the low half\", \"rep\": %d, \"kernel_ms\": %.4f}\n", rep, time_kernel(hi, lo));
            fflush(stdout);
        }
        return 0;
    }
    if (argc > 2 && strcmp(argv[2], "sweep") == 0) {
        // ring / tile knobs of the LDS-DMA kernel on ONE arena, destination in the fast-writing half and in the other:
        // does any setting that lost in round 1's (placement-blind) sweeps win once the placement is held fixed?
        const size_t pad = ((bytes + (2u << 20) - 1) >> 21) << 21;
        char* a;
        CHECK_HIP(hipMalloc(&a, 2 * pad));
        double* lo = (double*)a; double* hi = (double*)(a + pad);
        struct { const char* rh; const char* ns; const char* pf; const char* pairs; } cfg[] = {
            {"16", "8", "3", "2"}, {"16", "4", "3", "2"}, {"16", "16", "3", "2"}, {"16", "8", "1", "2"}, {"16", "8", "2", "2"}, {"16", "8", "5", "2"},
            {"16", "16", "7", "2"}, {"32", "8", "3", "2"}, {"8", "8", "3", "2"}, {"16", "8", "3", "1"}, {"16", "4", "2", "1"}, {"16", "8", "3", "4"}, {"32", "4", "2", "2"}};
        for (int rep = 0; rep < trials; ++rep)
            for (auto& c : cfg) {
                setenv("PXL_REPROJECT_RH", c.rh, 1); setenv("PXL_REPROJECT_NS", c.ns, 1); setenv("PXL_REPROJECT_PF", c.pf, 1); setenv("PXL_REPROJECT_PAIRS", c.pairs, 1);
                pxl_reproject_plan* pl;
                CHECK_PXL(pxl_reproject_plan_create(&win, shape_in, 0, NY, &wout, shape_out, 0, NY, &pl));
                g_plan = pl;
                printf("{\"how\": \"knob sweep\", \"rep\": %d, \"RH\": %s, \"NS\": %s, \"PF\": %s, \"PAIRS\": %s, \"dst_above_ms\": %.4f, \"dst_below_ms\": %.4f}\n",
                       rep, c.rh, c.ns, c.pf, c.pairs, time_kernel(lo, hi), time_kernel(hi, lo));
                fflush(stdout);
                pxl_reproject_plan_destroy(pl);
            }
        return 0;
    }
    for (int t = 0; t < trials; ++t) {
        {   // two hipMalloc calls, source first (what bench.py / torch do)
            double *src, *dst;
            CHECK_HIP(hipMalloc(&src, bytes)); CHECK_HIP(hipMalloc(&dst, bytes));
            printf("{\"how\": \"malloc src first\", \"trial\": %d, \"kernel_ms\": %.4f, \"dst_minus_src_MiB\": %.1f}\n", t, time_kernel(src, dst),
                   ((char*)dst - (char*)src) / 1048576.0); fflush(stdout);
            // the SAME two physical allocations with the roles swapped
            printf("{\"how\": \"same buffers, roles swapped\", \"trial\": %d, \"kernel_ms\": %.4f, \"dst_minus_src_MiB\": %.1f}\n", t, time_kernel(dst, src),
                   ((char*)src - (char*)dst) / 1048576.0); fflush(stdout);
            CHECK_HIP(hipFree(src)); CHECK_HIP(hipFree(dst));
        }
        {   // destination first
            double *src, *dst;
            CHECK_HIP(hipMalloc(&dst, bytes)); CHECK_HIP(hipMalloc(&src, bytes));
            printf("{\"how\": \"malloc dst first\", \"trial\": %d, \"kernel_ms\": %.4f, \"dst_minus_src_MiB\": %.1f}\n", t, time_kernel(src, dst),
                   ((char*)dst - (char*)src) / 1048576.0); fflush(stdout);
            CHECK_HIP(hipFree(src)); CHECK_HIP(hipFree(dst));
        }
        {   // one allocation, sub-allocated both ways
            const size_t pad = ((bytes + (2u << 20) - 1) >> 21) << 21;
            char* a;
            CHECK_HIP(hipMalloc(&a, 2 * pad));
            double* lo = (double*)a; double* hi = (double*)(a + pad);
            printf("{\"how\": \"arena addresses\", \"lo_va_mod_32GiB_in_GiB\": %.3f, \"hi_va_mod_32GiB_in_GiB\": %.3f}\n",
                   fmod((double)(uintptr_t)lo, 34359738368.0) / 1073741824.0, fmod((double)(uintptr_t)hi, 34359738368.0) / 1073741824.0);
            printf("{\"how\": \"arena, dst above src\", \"trial\": %d, \"kernel_ms\": %.4f}\n", t, time_kernel(lo, hi)); fflush(stdout);
            printf("{\"how\": \"arena, dst below src\", \"trial\": %d, \"kernel_ms\": %.4f}\n", t, time_kernel(hi, lo)); fflush(stdout);
            g_plan = g_plan_loads;
            printf("{\"how\": \"arena, LOADS ONLY from the low half\", \"trial\": %d, \"kernel_ms\": %.4f}\n", t, time_kernel(lo, hi));
            printf("{\"how\": \"arena, LOADS ONLY from the high half\", \"trial\": %d, \"kernel_ms\": %.4f}\n", t, time_kernel(hi, lo));
            g_plan = g_plan_stores;
            printf("{\"how\": \"arena, STORES ONLY to the high half\", \"trial\": %d, \"kernel_ms\": %.4f}\n", t, time_kernel(lo, hi));
            printf("{\"how\": \"arena, STORES ONLY to the low half\", \"trial\": %d, \"kernel_ms\": %.4f}\n", t, time_kernel(hi, lo));
            for (int o = 0; o < 2; ++o) {
                g_plan = g_plan_st_ord[o];
                printf("{\"how\": \"arena, STORES ONLY, tile order %d\", \"trial\": %d, \"to_high_ms\": %.4f, \"to_low_ms\": %.4f}\n", o + 1, t, time_kernel(lo, hi), time_kernel(hi, lo));
                g_plan = g_plan_ord[o];
                printf("{\"how\": \"arena, full kernel, tile order %d\", \"trial\": %d, \"dst_above_ms\": %.4f, \"dst_below_ms\": %.4f}\n", o + 1, t, time_kernel(lo, hi), time_kernel(hi, lo));
            }
            g_plan = g_plan_st_rr;
            printf("{\"how\": \"arena, STORES ONLY, one front (tile = block)\", \"trial\": %d, \"to_high_ms\": %.4f, \"to_low_ms\": %.4f}\n", t, time_kernel(lo, hi), time_kernel(hi, lo));
            for (int k = 0; k < 5; ++k) {
                g_plan = g_plan_st_chunk[k];
                printf("{\"how\": \"arena, STORES ONLY, XCD pieces of %d tiles\", \"trial\": %d, \"to_high_ms\": %.4f, \"to_low_ms\": %.4f}\n", kChunks[k], t,
                       time_kernel(lo, hi), time_kernel(hi, lo));
            }
            for (int k = 0; k < 5; ++k) {
                g_plan = g_plan_chunk[k];
                printf("{\"how\": \"arena, XCD pieces of %d tiles\", \"trial\": %d, \"dst_above_ms\": %.4f, \"dst_below_ms\": %.4f}\n", kChunks[k], t,
                       time_kernel(lo, hi), time_kernel(hi, lo));
            }
            g_plan = g_plan_full; fflush(stdout);
            CHECK_HIP(hipFree(a));
        }
        {   // where in a big allocation does the store rate change?  One 70 GB allocation, the destination at growing offsets
            const size_t big = (size_t)70 << 30;
            char* a;
            CHECK_HIP(hipMalloc(&a, big));
            g_plan = g_plan_stores;
            const double offs_gib[] = {0, 4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 44, 48};
            const double G32 = 34359738368.0;
            for (double og : offs_gib) {
                const size_t off = (size_t)(og * 1024) << 20;
                if (off + bytes > big) break;
                const double va = (double)(uintptr_t)(a + off);
                printf("{\"how\": \"70 GB allocation, STORES ONLY at offset\", \"trial\": %d, \"offset_GiB\": %.2f, \"dst_va_mod_32GiB_in_GiB\": %.3f, "
                       "\"dst_end_mod_32GiB_in_GiB\": %.3f, \"kernel_ms\": %.4f}\n", t, og, fmod(va, G32) / 1073741824.0,
                       fmod(va + (double)bytes, G32) / 1073741824.0,
                       time_kernel((double*)(a + (off + bytes <= big / 2 ? big - bytes : 0)), (double*)(a + off)));
                fflush(stdout);
            }
            g_plan = g_plan_full;
            CHECK_HIP(hipFree(a));
        }
        if (t == 0) {
            VmmBuf sb = vmm_alloc(bytes, (size_t)2 << 20), db = vmm_alloc(bytes, (size_t)2 << 20);
            printf("{\"how\": \"vmm 2 MiB handles\", \"trial\": %d, \"handles_per_map\": %zu, \"kernel_ms\": %.4f}\n", t, sb.handles.size(),
                   time_kernel((double*)sb.va, (double*)db.va));
            printf("{\"how\": \"vmm 2 MiB handles, roles swapped\", \"trial\": %d, \"kernel_ms\": %.4f}\n", t, time_kernel((double*)db.va, (double*)sb.va));
            fflush(stdout);
            vmm_free(sb); vmm_free(db);
        }
    }
    pxl_reproject_plan_destroy(g_plan_full); pxl_reproject_plan_destroy(g_plan_loads); pxl_reproject_plan_destroy(g_plan_stores);
    return 0;
}
