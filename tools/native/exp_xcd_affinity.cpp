// exp_xcd_affinity.cpp -- is device memory equally far from every XCD?  One big allocation; for windows spaced through it and for
// each XCD v (workgroups with blockIdx % 8 == v run on one XCD; the others exit at once): the rate of a plain 16-byte-per-lane
// store stream of that XCD alone into the window; then all eight XCDs into ONE window, and into EIGHT windows (XCD v -> window
// (v + shift) % 8) -- the shape of the reprojection's eight write fronts.  Prints one JSON line per measurement.
//   hipcc --offload-arch=gfx950 -O2 tools/native/exp_xcd_affinity.cpp -o tools/native/exp_xcd_affinity && tools/native/exp_xcd_affinity 224 32
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

// XCD v writes `bytes` bytes starting at base[v] (its own window); blocks of other XCDs exit when only == v is asked for
__global__ __launch_bounds__(256) void k_store(char* const* base, size_t bytes, int only) {
    const int v = blockIdx.x & 7;
    if (only >= 0 && v != only) return;
    const size_t j = blockIdx.x >> 3, nj = gridDim.x >> 3;
    uint4* p = reinterpret_cast<uint4*>(base[v]);
    const size_t n = bytes / 16;
    const uint4 val = make_uint4(1, 2, 3, 4);
    // contiguous 64 KiB pieces per block, walking upwards: a moving front like the reprojection's
    const size_t piece = 4096;                                           // uint4 per piece
    for (size_t q = j; q * piece < n; q += nj)
        for (size_t i = q * piece + threadIdx.x; i < (q + 1) * piece && i < n; i += 256) p[i] = val;
}

static hipStream_t st;
static float run(char* const* dbase, size_t bytes, int only, int reps = 5) {
    std::vector<float> t(reps);
    hipEvent_t e0, e1;
    CHECK_HIP(hipEventCreate(&e0)); CHECK_HIP(hipEventCreate(&e1));
    const int grid = 8 * 256 * 4;
    hipLaunchKernelGGL(k_store, dim3(grid), dim3(256), 0, st, dbase, bytes, only);
    for (int r = 0; r < reps; ++r) {
        CHECK_HIP(hipEventRecord(e0, st));
        hipLaunchKernelGGL(k_store, dim3(grid), dim3(256), 0, st, dbase, bytes, only);
        CHECK_HIP(hipEventRecord(e1, st));
        CHECK_HIP(hipEventSynchronize(e1));
        CHECK_HIP(hipEventElapsedTime(&t[r], e0, e1));
    }
    std::sort(t.begin(), t.end());
    return t[reps / 2];
}

int main(int argc, char** argv) {
    const size_t GiB = 1ull << 30;
    const size_t total = (size_t)(argc > 1 ? atoi(argv[1]) : 224) * GiB;
    const size_t stride = (size_t)(argc > 2 ? atoi(argv[2]) : 32) * GiB;
    CHECK_HIP(hipStreamCreate(&st));
    char* arena;
    CHECK_HIP(hipMalloc(&arena, total));
    char** dbase;
    CHECK_HIP(hipMalloc(&dbase, 8 * sizeof(char*)));
    auto set = [&](const size_t* off) {
        char* h[8];
        for (int v = 0; v < 8; ++v) h[v] = arena + off[v];
        CHECK_HIP(hipMemcpy(dbase, h, sizeof h, hipMemcpyHostToDevice));
    };
    if (argc > 3 && !strcmp(argv[3], "classes")) {
        // which parts of the allocation behave like "different regions"?  Windows of 1 GiB every `stride` GiB; for every pair
        // (a, b): XCDs 0, 2, 4, 6 write four 256 MiB pieces of window a, XCDs 1, 3, 5, 7 of window b (a == b: all eight
        // pieces in the one 2 GiB window).  Fast (7 TB/s) = the two windows differ in whatever the boundary effect is about.
        const size_t piece = 256ull << 20;
        std::vector<size_t> wins;
        for (size_t w = 0; w + 2 * GiB <= total; w += stride) wins.push_back(w);
        printf("{\"windows_GiB\": [");
        for (size_t i = 0; i < wins.size(); ++i) printf("%s%zu", i ? ", " : "", wins[i] / GiB);
        printf("]}\n");
        for (size_t a = 0; a < wins.size(); ++a) {
            printf("{\"a_GiB\": %zu, \"GBs_vs_b\": [", wins[a] / GiB);
            for (size_t b = 0; b < wins.size(); ++b) {
                size_t off[8];
                for (int v = 0; v < 8; ++v) off[v] = (a == b) ? wins[a] + v * piece : ((v & 1) ? wins[b] : wins[a]) + (v / 2) * piece;
                set(off);
                printf("%s%.0f", b ? ", " : "", 8.0 * piece / 1e6 / run(dbase, piece, -1, 3));
            }
            printf("]}\n");
            fflush(stdout);
        }
        return 0;
    }
    const size_t W = 1 * GiB;
    // 1. one XCD at a time into one window
    for (size_t w = 0; w + W <= total; w += stride) {
        size_t off[8];
        for (int v = 0; v < 8; ++v) off[v] = w;
        set(off);
        printf("{\"test\": \"one XCD -> window at %zu GiB\", \"GBs_by_xcd\": [", w / GiB);
        for (int v = 0; v < 8; ++v) printf("%s%.0f", v ? ", " : "", W / 1e6 / run(dbase, W, v));
        printf("]}\n");
        fflush(stdout);
    }
    // 2. eight fronts: XCD v -> piece v of a 7 GiB destination starting at `w` (the reprojection's shape), sliding w
    const size_t D = 7 * GiB, piece = D / 8;
    for (size_t w = 0; w + D <= total; w += 4 * GiB) {
        size_t off[8];
        for (int v = 0; v < 8; ++v) off[v] = w + v * piece;
        set(off);
        float a = run(dbase, piece, -1);
        for (int v = 0; v < 8; ++v) off[v] = w + ((v + 3) % 8) * piece;        // the same pieces, dealt to other XCDs
        set(off);
        float b = run(dbase, piece, -1);
        for (int v = 0; v < 8; ++v) off[v] = w;                                  // one front: everybody in the first piece
        set(off);
        float c = run(dbase, piece, -1);
        printf("{\"test\": \"eight fronts in a 7 GiB destination at %zu GiB\", \"GBs\": %.0f, \"GBs_pieces_rotated_by_3\": %.0f, \"GBs_all_XCDs_in_one_piece_8x\": %.0f}\n",
               w / GiB, D / 1e6 / a, D / 1e6 / b, 8.0 * piece / 1e6 / c);
        fflush(stdout);
    }
    return 0;
}
