// exp_vmm_classes.cpp -- what distinguishes the physical regions in which the reprojection's destination is fast?
// Round 3: sliding the destination through one 220 GiB allocation (tools/research/exp_scan_placement.py) shows the kernel's stores
// running at 7.0 instead of 6.0 TB/s exactly when the destination STRADDLES a boundary between two regions of the
// allocation (boundaries every 32 GiB, denser near the allocation's end), fastest with the boundary in the middle.
// This host builds the destination out of separately created physical handles (hipMemCreate / hipMemMap) instead:
//   scan    destination = 'w' consecutive handles of the pool starting at handle i (the allocation-order picture)
//   pairs   destination = handles alternating between a run starting at i and a run starting at j (which (i, j) pairs
//           behave like a straddling destination?)
//   policy  destination = alternating handles of two pools created with a ballast of G GiB between them (what an
//           allocator could do without probing anything), against the plain order
// Workload: the 1' -> 0.5' refinement (config 3), stores only and the full launch.  One JSON line per measurement.
//   hipcc --offload-arch=gfx950 -O2 -I include tools/research/exp_vmm_classes.cpp -L pixell.jl_amd -lpixell_hip \
//         -Wl,-rpath,$PWD/pixell.jl_amd -o tools/research/exp_vmm_classes && tools/research/exp_vmm_classes scan 400
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "pixell_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
#define CHECK_PXL(x) do { int rc_ = (x); if (rc_ != 0) { char m_[256]; pxl_last_error(m_, sizeof m_); fprintf(stderr, "%s -> %d: %s\n", #x, rc_, m_); exit(3); } } while (0)

static int64_t NX = 21600, NY = 10801, NXO = 43200, NYO = 21601, NC = 1;
static pxl_reproject_plan *g_full, *g_stores;
static hipStream_t g_st;
static const size_t H = 512ull << 20;                  // handle size (1 GiB handles are refused by hipMemSetAccess here)
static hipMemAllocationProp g_prop;

static double time_plan(pxl_reproject_plan* pl, double* src, double* dst, int reps = 5) {
    CHECK_PXL(pxl_reproject_build_tables(pl, g_st));
    for (int k = 0; k < 2; ++k) CHECK_PXL(pxl_reproject_execute_rows(pl, src, dst, 0, NYO, g_st));
    std::vector<float> t(reps);
    hipEvent_t e0, e1;
    CHECK_HIP(hipEventCreate(&e0)); CHECK_HIP(hipEventCreate(&e1));
    for (int r = 0; r < reps; ++r) {
        CHECK_HIP(hipEventRecord(e0, g_st));
        CHECK_PXL(pxl_reproject_execute_rows(pl, src, dst, 0, NYO, g_st));
        CHECK_HIP(hipEventRecord(e1, g_st));
        CHECK_HIP(hipEventSynchronize(e1));
        CHECK_HIP(hipEventElapsedTime(&t[r], e0, e1));
    }
    CHECK_HIP(hipEventDestroy(e0)); CHECK_HIP(hipEventDestroy(e1));
    std::sort(t.begin(), t.end());
    return t[reps / 2];
}

static hipMemGenericAllocationHandle_t make_handle() {
    hipMemGenericAllocationHandle_t h;
    CHECK_HIP(hipMemCreate(&h, H, &g_prop, 0));
    return h;
}
static void map_list(void* va, const std::vector<hipMemGenericAllocationHandle_t>& hs) {
    for (size_t k = 0; k < hs.size(); ++k) CHECK_HIP(hipMemMap((char*)va + k * H, H, 0, hs[k], 0));
    hipMemAccessDesc acc = {};
    acc.location = g_prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CHECK_HIP(hipMemSetAccess(va, hs.size() * H, &acc, 1));
}
static void unmap(void* va, size_t n) { CHECK_HIP(hipMemUnmap(va, n * H)); }

int main(int argc, char** argv) {
    const char* mode = argc > 1 ? argv[1] : "scan";
    const int npool = argc > 2 ? atoi(argv[2]) : 400;
    if (getenv("PXL_WL") && !strcmp(getenv("PXL_WL"), "cfg4")) { NX = 43200; NY = 21601; NXO = 43200; NYO = 21601; NC = 3; }
    const double pi = 3.141592653589793;
    pxl_car_wcs win = {{-360.0 / NX, 180.0 / (NY - 1)}, {floor(NX / 2.0) + 0.5, (NY + 1) / 2.0}, {(2 * pi / NX) * 90 / pi, 0.0}, pi / 180};
    pxl_car_wcs wout = {{-360.0 / NXO, 180.0 / (NYO - 1)}, {floor(NXO / 2.0) + 0.5, (NYO + 1) / 2.0}, {(2 * pi / NXO) * 90 / pi, 0.0}, pi / 180};
    if (NXO == NX) { wout.crpix[0] += 0.5; wout.crpix[1] += 0.5; }
    const int64_t shape_in[3] = {NX, NY, NC}, shape_out[2] = {NXO, NYO};
    CHECK_HIP(hipStreamCreate(&g_st));
    CHECK_PXL(pxl_reproject_plan_create(&win, shape_in, 0, NY, &wout, shape_out, 0, NYO, &g_full));
    setenv("PXL_REPROJECT_FLAGS", "64", 1);
    CHECK_PXL(pxl_reproject_plan_create(&win, shape_in, 0, NY, &wout, shape_out, 0, NYO, &g_stores));
    unsetenv("PXL_REPROJECT_FLAGS");
    memset(&g_prop, 0, sizeof g_prop);
    g_prop.type = hipMemAllocationTypePinned;
    g_prop.location.type = hipMemLocationTypeDevice;
    g_prop.location.id = 0;
    const size_t src_bytes = (size_t)NX * NY * NC * 8, dst_bytes = (size_t)NXO * NYO * NC * 8;
    const int ws = (int)((src_bytes + H - 1) / H), wd = (int)((dst_bytes + H - 1) / H);
    void *src_va, *dst_va;
    CHECK_HIP(hipMemAddressReserve(&src_va, (size_t)ws * H, 0, nullptr, 0));
    CHECK_HIP(hipMemAddressReserve(&dst_va, (size_t)wd * H, 0, nullptr, 0));
    printf("{\"mode\": \"%s\", \"handle_MiB\": %zu, \"src_handles\": %d, \"dst_handles\": %d, \"pool_handles\": %d, \"src_va\": \"%p\", \"dst_va\": \"%p\", \"dst_va_mod_32GiB_in_GiB\": %.3f}\n",
           mode, H >> 20, ws, wd, npool, src_va, dst_va, fmod((double)(uintptr_t)dst_va, 34359738368.0) / 1073741824.0);

    if (!strcmp(mode, "gran")) {
        // handle size: the destination (and the source) built from handles of 2 MiB ... 512 MiB, created and mapped in order;
        // is hipMemCreate lazy (free memory before / after)?  `npool` repetitions of the whole series in this process.
        const size_t sizes[] = {512ull << 20, 64ull << 20, 8ull << 20, 2ull << 20};
        for (int rep = 0; rep < npool; ++rep)
            for (size_t hs : sizes) {
                size_t free0, free1, free2, tot;
                CHECK_HIP(hipMemGetInfo(&free0, &tot));
                const size_t nd = ((size_t)wd * H + hs - 1) / hs, nsrc = ((size_t)ws * H + hs - 1) / hs;
                std::vector<hipMemGenericAllocationHandle_t> D(nd), S2(nsrc);
                for (auto& h : S2) CHECK_HIP(hipMemCreate(&h, hs, &g_prop, 0));
                for (auto& h : D) CHECK_HIP(hipMemCreate(&h, hs, &g_prop, 0));
                CHECK_HIP(hipMemGetInfo(&free1, &tot));
                hipMemAccessDesc acc = {};
                acc.location = g_prop.location;
                acc.flags = hipMemAccessFlagsProtReadWrite;
                for (size_t k = 0; k < nsrc; ++k) CHECK_HIP(hipMemMap((char*)src_va + k * hs, hs, 0, S2[k], 0));
                CHECK_HIP(hipMemSetAccess(src_va, nsrc * hs, &acc, 1));
                for (size_t k = 0; k < nd; ++k) CHECK_HIP(hipMemMap((char*)dst_va + k * hs, hs, 0, D[k], 0));
                CHECK_HIP(hipMemSetAccess(dst_va, nd * hs, &acc, 1));
                CHECK_HIP(hipMemGetInfo(&free2, &tot));
                CHECK_PXL(pxl_fill_random_f64((double*)src_va, NX * NY * NC, 1234, 0, 0, g_st));
                double s = time_plan(g_stores, (double*)src_va, (double*)dst_va), f = time_plan(g_full, (double*)src_va, (double*)dst_va);
                printf("{\"rep\": %d, \"handle_MiB\": %zu, \"handles\": %zu, \"stores_only_ms\": %.4f, \"full_ms\": %.4f, \"free_GiB_before_create_after_create_after_map\": [%.2f, %.2f, %.2f]}\n",
                       rep, hs >> 20, nd, s, f, free0 / 1073741824.0, free1 / 1073741824.0, free2 / 1073741824.0);
                fflush(stdout);
                CHECK_HIP(hipMemUnmap(dst_va, nd * hs));
                CHECK_HIP(hipMemUnmap(src_va, nsrc * hs));
                for (auto h : D) CHECK_HIP(hipMemRelease(h));
                for (auto h : S2) CHECK_HIP(hipMemRelease(h));
            }
        return 0;
    }
    if (!strcmp(mode, "policy")) {
        // an allocator's view: no pool, no probing.  A = wd/2 handles, ballast of G GiB, B = the other half; the destination
        // takes A and B alternately ("interleaved") or A then B ("split": one boundary in the middle) or is created in one go
        // ("plain").  The source is created last.  Repeated `npool` times in this process (fresh handles every time).
        const int G = argc > 3 ? atoi(argv[3]) : 64;
        for (int trial = 0; trial < npool; ++trial) {
            std::vector<hipMemGenericAllocationHandle_t> A, B, ballast, S, P;
            for (int k = 0; k < (wd + 1) / 2; ++k) A.push_back(make_handle());
            for (int k = 0; k < G * 2; ++k) ballast.push_back(make_handle());
            for (int k = 0; k < wd / 2; ++k) B.push_back(make_handle());
            for (auto h : ballast) CHECK_HIP(hipMemRelease(h));
            for (int k = 0; k < ws; ++k) S.push_back(make_handle());
            map_list(src_va, S);
            CHECK_PXL(pxl_fill_random_f64((double*)src_va, NX * NY * NC, 1234, 0, 0, g_st));
            std::vector<hipMemGenericAllocationHandle_t> inter, split;
            for (int k = 0; k < wd; ++k) inter.push_back((k & 1) ? B[k / 2] : A[k / 2]);
            for (auto h : A) split.push_back(h);
            for (auto h : B) split.push_back(h);
            map_list(dst_va, inter);
            double f1 = time_plan(g_full, (double*)src_va, (double*)dst_va), s1 = time_plan(g_stores, (double*)src_va, (double*)dst_va);
            unmap(dst_va, wd);
            map_list(dst_va, split);
            double f2 = time_plan(g_full, (double*)src_va, (double*)dst_va), s2 = time_plan(g_stores, (double*)src_va, (double*)dst_va);
            unmap(dst_va, wd);
            for (int k = 0; k < wd; ++k) P.push_back(make_handle());          // a plain destination, created in one go
            map_list(dst_va, P);
            double f3 = time_plan(g_full, (double*)src_va, (double*)dst_va), s3 = time_plan(g_stores, (double*)src_va, (double*)dst_va);
            unmap(dst_va, wd);
            unmap(src_va, ws);
            printf("{\"trial\": %d, \"ballast_GiB\": %d, \"interleaved\": [%.4f, %.4f], \"split\": [%.4f, %.4f], \"plain\": [%.4f, %.4f], \"order\": \"[full_ms, stores_only_ms]\"}\n",
                   trial, G, f1, s1, f2, s2, f3, s3);
            fflush(stdout);
            for (auto h : A) CHECK_HIP(hipMemRelease(h));
            for (auto h : B) CHECK_HIP(hipMemRelease(h));
            for (auto h : S) CHECK_HIP(hipMemRelease(h));
            for (auto h : P) CHECK_HIP(hipMemRelease(h));
        }
        return 0;
    }

    std::vector<hipMemGenericAllocationHandle_t> pool;
    for (int k = 0; k < npool; ++k) pool.push_back(make_handle());
    // the source lives on the last handles of the pool
    std::vector<hipMemGenericAllocationHandle_t> S(pool.end() - ws, pool.end());
    map_list(src_va, S);
    CHECK_PXL(pxl_fill_random_f64((double*)src_va, NX * NY * NC, 1234, 0, 0, g_st));
    const int usable = npool - ws;
    if (!strcmp(mode, "scan")) {
        const int step = argc > 3 ? atoi(argv[3]) : 2;
        for (int i = 0; i + wd <= usable; i += step) {
            std::vector<hipMemGenericAllocationHandle_t> hs(pool.begin() + i, pool.begin() + i + wd);
            map_list(dst_va, hs);
            double f, s;
            if (getenv("PXL_ORDER") && !strcmp(getenv("PXL_ORDER"), "sf")) { s = time_plan(g_stores, (double*)src_va, (double*)dst_va); f = time_plan(g_full, (double*)src_va, (double*)dst_va); }
            else { f = time_plan(g_full, (double*)src_va, (double*)dst_va); s = time_plan(g_stores, (double*)src_va, (double*)dst_va); }
            unmap(dst_va, wd);
            printf("{\"first_handle\": %d, \"offset_GiB\": %.1f, \"full_ms\": %.4f, \"stores_only_ms\": %.4f}\n", i, i * (double)H / (1 << 30), f, s);
            fflush(stdout);
        }
    } else if (!strcmp(mode, "va")) {
        // the SAME physical handles (the first wd of the pool) mapped at different VIRTUAL addresses inside one big reservation:
        // is the fast / slow distinction a property of the virtual address?
        const int span = argc > 3 ? atoi(argv[3]) : 96;          // GiB of address space
        const double vstep = argc > 4 ? atof(argv[4]) : 1.0;     // GiB between positions (multiples of the handle size)
        void* big;
        CHECK_HIP(hipMemAddressReserve(&big, (size_t)span << 30, 0, nullptr, 0));
        printf("{\"va_base\": \"%p\", \"va_base_mod_32GiB_in_GiB\": %.3f, \"src_va\": \"%p\"}\n", big,
               fmod((double)(uintptr_t)big, 34359738368.0) / 1073741824.0, src_va);
        std::vector<hipMemGenericAllocationHandle_t> hs(pool.begin(), pool.begin() + wd);
        for (double off = 0; off + wd * 0.5 <= span; off += vstep) {
            char* va = (char*)big + (size_t)(off * 2 + 0.5) * H;
            map_list(va, hs);
            double s = time_plan(g_stores, (double*)src_va, (double*)va, 3), f = time_plan(g_full, (double*)src_va, (double*)va, 3);
            CHECK_HIP(hipMemUnmap(va, (size_t)wd * H));
            printf("{\"va_offset_GiB\": %.1f, \"va_mod_32GiB_in_GiB\": %.2f, \"stores_only_ms\": %.4f, \"full_ms\": %.4f}\n", off,
                   fmod((double)(uintptr_t)va, 34359738368.0) / 1073741824.0, s, f);
            fflush(stdout);
        }
    } else if (!strcmp(mode, "align")) {
        // the same physical handles mapped at virtual addresses of different ALIGNMENT (the driver can only use page-table
        // fragments as large as the alignment the virtual and the physical address share)
        void* big;
        CHECK_HIP(hipMemAddressReserve(&big, 64ull << 30, 1ull << 30, nullptr, 0));        // 1 GiB aligned
        printf("{\"va_base\": \"%p\"}\n", big);
        std::vector<hipMemGenericAllocationHandle_t> hs(pool.begin(), pool.begin() + wd);
        const size_t offs[] = {0, 2ull << 20, 4ull << 20, 32ull << 20, 256ull << 20, 512ull << 20, 1ull << 30, (1ull << 30) + (2ull << 20), 0};
        for (int rep = 0; rep < 2; ++rep)
            for (size_t off : offs) {
                char* va = (char*)big + (8ull << 30) + off;
                map_list(va, hs);
                double s = time_plan(g_stores, (double*)src_va, (double*)va, 5), f = time_plan(g_full, (double*)src_va, (double*)va, 5);
                CHECK_HIP(hipMemUnmap(va, (size_t)wd * H));
                printf("{\"va_offset_MiB_from_a_1GiB_boundary\": %zu, \"stores_only_ms\": %.4f, \"full_ms\": %.4f}\n", off >> 20, s, f);
                fflush(stdout);
            }
    } else if (!strcmp(mode, "pairs")) {
        // destination = alternating handles of the runs starting at i and at j; i, j on a grid of `step` handles
        const int step = argc > 3 ? atoi(argv[3]) : 32;
        const int half = (wd + 1) / 2;
        for (int i = 0; i + half <= usable; i += step)
            for (int j = i; j + half <= usable; j += step) {
                if (j < i + half && j != i) continue;
                std::vector<hipMemGenericAllocationHandle_t> hs;
                if (i == j) { if (i + wd > usable) continue; hs.assign(pool.begin() + i, pool.begin() + i + wd); }
                else for (int k = 0; k < wd; ++k) hs.push_back((k & 1) ? pool[j + k / 2] : pool[i + k / 2]);
                map_list(dst_va, hs);
                double s = time_plan(g_stores, (double*)src_va, (double*)dst_va, 3), f = time_plan(g_full, (double*)src_va, (double*)dst_va, 3);
                unmap(dst_va, wd);
                printf("{\"i\": %d, \"j\": %d, \"i_GiB\": %.0f, \"j_GiB\": %.0f, \"stores_only_ms\": %.4f, \"full_ms\": %.4f}\n", i, j, i * (double)H / (1 << 30), j * (double)H / (1 << 30), s, f);
                fflush(stdout);
            }
    }
    return 0;
}
