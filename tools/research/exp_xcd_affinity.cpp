// exp_xcd_affinity.cpp -- is device memory equally far from every XCD?  One big allocation; for windows spaced through it and for
// each XCD v (workgroups with blockIdx % 8 == v run on one XCD; the others exit at once): the rate of a plain 16-byte-per-lane
// store stream of that XCD alone into the window; then all eight XCDs into ONE window, and into EIGHT windows (XCD v -> window
// (v + shift) % 8) -- the shape of the reprojection's eight write fronts.  Prints one JSON line per measurement.
//   hipcc --offload-arch=gfx950 -O2 tools/research/exp_xcd_affinity.cpp -o tools/research/exp_xcd_affinity && tools/research/exp_xcd_affinity 224 32
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

// F fronts in one destination: front f = (XCD v, sub-front s) for F >= 8 (each XCD deals its blocks round-robin over its F / 8 pieces);
// for F < 8 the XCDs v, v + F, ... share front v % F (their blocks interleave inside the piece)
__global__ __launch_bounds__(256) void k_fronts(char* base, size_t total_bytes, int F) {
    const int v = blockIdx.x & 7;
    const size_t j = blockIdx.x >> 3, nj = gridDim.x >> 3;
    const size_t piece_bytes = total_bytes / F;
    const uint4 val = make_uint4(1, 2, 3, 4);
    const size_t chunk = 4096;                          // uint4 per 64 KiB step
    const size_t n = piece_bytes / 16;
    if (F >= 8) {
        const int per = F / 8;
        const int s_ = (int)(j % per);
        uint4* p = reinterpret_cast<uint4*>(base + (size_t)(v * per + s_) * piece_bytes);
        for (size_t q = j / per; q * chunk < n; q += nj / per)
            for (size_t i = q * chunk + threadIdx.x; i < (q + 1) * chunk && i < n; i += 256) p[i] = val;
    } else {
        const int share = 8 / F;                        // XCDs per front
        uint4* p = reinterpret_cast<uint4*>(base + (size_t)(v % F) * piece_bytes);
        const size_t jj = j * share + v / F, njj = nj * share;
        for (size_t q = jj; q * chunk < n; q += njj)
            for (size_t i = q * chunk + threadIdx.x; i < (q + 1) * chunk && i < n; i += 256) p[i] = val;
    }
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
    if (argc > 3 && !strcmp(argv[3], "fronts")) {
        // how many write fronts should a kernel keep inside ONE class?  8 GiB at the start of the allocation (one block, one class)
        const size_t D = 8 * GiB;
        for (int rep = 0; rep < 2; ++rep)
            for (int F : {1, 2, 4, 8, 16, 32, 64, 128, 512}) {
                std::vector<float> t(5);
                hipEvent_t e0, e1;
                CHECK_HIP(hipEventCreate(&e0)); CHECK_HIP(hipEventCreate(&e1));
                for (int r = -1; r < 5; ++r) {
                    CHECK_HIP(hipEventRecord(e0, st));
                    hipLaunchKernelGGL(k_fronts, dim3(8 * 256 * 4), dim3(256), 0, st, arena, D, F);
                    CHECK_HIP(hipEventRecord(e1, st));
                    CHECK_HIP(hipEventSynchronize(e1));
                    if (r >= 0) CHECK_HIP(hipEventElapsedTime(&t[r], e0, e1));
                }
                std::sort(t.begin(), t.end());
                printf("{\"fronts\": %d, \"GBs\": %.0f}\n", F, D / 1e6 / t[2]);
                fflush(stdout);
            }
        return 0;
    }
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
