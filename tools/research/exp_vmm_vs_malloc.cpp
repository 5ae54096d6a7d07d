// exp_vmm_vs_malloc.cpp -- in ONE process: (1) a hipMalloc'ed arena, 1 GiB windows every 16 GiB, the eight-front store probe for every
// pair of windows (four fronts in each); (2) the same probe between 1 GiB groups of hipMemCreate handles (group k against group 0
// and against the previous group).  Does a process in which every VMM group looks alike still see classes in a hipMalloc arena?
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
__global__ __launch_bounds__(256) void k_probe(char* a, char* b, size_t piece_bytes) {
    const int v = blockIdx.x & 7;
    const size_t j = blockIdx.x >> 3, nj = gridDim.x >> 3;
    uint4* p = reinterpret_cast<uint4*>(((v & 1) ? b : a) + (size_t)(v >> 1) * piece_bytes);
    const size_t n = piece_bytes / 16, chunk = 4096;
    const uint4 val = make_uint4(0, 0, 0, 0);
    for (size_t q = j; q * chunk < n; q += nj)
        for (size_t i = q * chunk + threadIdx.x; i < (q + 1) * chunk && i < n; i += 256) p[i] = val;
}
static hipStream_t st;
static char* g_arena;
static std::vector<size_t> g_w;
static float probe(char* a, char* b) {
    const size_t piece = 256ull << 20;
    float t[3];
    hipEvent_t e0, e1;
    CHECK_HIP(hipEventCreate(&e0)); CHECK_HIP(hipEventCreate(&e1));
    for (int r = -1; r < 3; ++r) {
        CHECK_HIP(hipEventRecord(e0, st));
        hipLaunchKernelGGL(k_probe, dim3(8 * 256 * 2), dim3(256), 0, st, a, b, piece);
        CHECK_HIP(hipEventRecord(e1, st));
        CHECK_HIP(hipEventSynchronize(e1));
        if (r >= 0) CHECK_HIP(hipEventElapsedTime(&t[r], e0, e1));
    }
    CHECK_HIP(hipEventDestroy(e0)); CHECK_HIP(hipEventDestroy(e1));
    std::sort(t, t + 3);
    return t[1] * 1e3f;
}
int main(int argc, char** argv) {
    const size_t GiB = 1ull << 30, H = 512ull << 20;
    const int arena_gib = argc > 1 ? atoi(argv[1]) : 128, ngroups = argc > 2 ? atoi(argv[2]) : 200;
    CHECK_HIP(hipStreamCreate(&st));
    {
        char* arena;
        CHECK_HIP(hipMalloc(&arena, (size_t)arena_gib * GiB));
        std::vector<size_t> w;
        for (size_t o = 0; o + GiB <= (size_t)arena_gib * GiB; o += 16 * GiB) w.push_back(o);
        printf("{\"hipMalloc_arena_GiB\": %d, \"probe_us_rows\": [", arena_gib);
        for (size_t a = 0; a < w.size(); ++a) {
            printf("%s[", a ? ", " : "");
            for (size_t b = 0; b < w.size(); ++b) printf("%s%.0f", b ? ", " : "", a == b ? 0.f : probe(arena + w[a], arena + w[b]));
            printf("]");
        }
        printf("]}\n");
        fflush(stdout);
        g_arena = arena; g_w = w;
    }
    {
        // (1b) many SEPARATE hipMalloc calls of 1 GiB each: block k against block 0 and against the previous one
        const int nb = argc > 3 ? atoi(argv[3]) : 0;
        std::vector<char*> blk(nb);
        for (auto& b : blk) CHECK_HIP(hipMalloc(&b, GiB));
        if (nb) printf("{\"separate_hipMalloc_1GiB_blocks\": %d, \"probe_us_vs_block0_and_vs_previous\": [", nb);
        for (int k = 1; k < nb; ++k) printf("%s[%.0f, %.0f]", k > 1 ? ", " : "", probe(blk[k], blk[0]), probe(blk[k], blk[k - 1]));
        if (nb) printf("]}\n");
        for (auto b : blk) CHECK_HIP(hipFree(b));
    }
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    std::vector<hipMemGenericAllocationHandle_t> hs(2 * ngroups);
    for (auto& h : hs) CHECK_HIP(hipMemCreate(&h, H, &prop, 0));
    char *va, *vb;
    CHECK_HIP(hipMemAddressReserve((void**)&va, GiB, 0, nullptr, 0));
    CHECK_HIP(hipMemAddressReserve((void**)&vb, GiB, 0, nullptr, 0));
    auto mapg = [&](char* v, int g) {
        for (int k = 0; k < 2; ++k) CHECK_HIP(hipMemMap(v + k * H, H, 0, hs[2 * g + k], 0));
        CHECK_HIP(hipMemSetAccess(v, GiB, &acc, 1));
    };
    // (2a) every 4th VMM group against every window of the (still allocated) hipMalloc arena
    printf("{\"vmm_group_vs_arena_windows_us\": [");
    for (int g = 0; g < ngroups; g += 4) {
        mapg(va, g);
        printf("%s[", g ? ", " : "");
        for (size_t b = 0; b < g_w.size(); ++b) printf("%s%.0f", b ? ", " : "", probe(va, g_arena + g_w[b]));
        printf("]");
        CHECK_HIP(hipMemUnmap(va, GiB));
    }
    printf("]}\n");
    fflush(stdout);
    printf("{\"vmm_groups\": %d, \"probe_us_vs_group0_and_vs_previous\": [", ngroups);
    for (int g = 1; g < ngroups; ++g) {
        mapg(va, g); mapg(vb, 0);
        const float t0 = probe(va, vb);
        CHECK_HIP(hipMemUnmap(vb, GiB));
        mapg(vb, g - 1);
        const float t1 = probe(va, vb);
        CHECK_HIP(hipMemUnmap(vb, GiB)); CHECK_HIP(hipMemUnmap(va, GiB));
        printf("%s[%.0f, %.0f]", g > 1 ? ", " : "", t0, t1);
    }
    printf("]}\n");
    return 0;
}
