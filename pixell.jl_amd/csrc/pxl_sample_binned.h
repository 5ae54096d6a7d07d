// pxl_sample_binned.h -- tile-binned scattered sampler (BASELINE config 5); included by pxl_kernels.hip (one
// translation unit, -ffp-contract=off).
//
// Why it exists: the direct gather (k_sample_bilinear) touches ~2.1 random 64-byte DRAM sectors per point on a
// multi-GB map; rocprofv3 (profiles/r02_cfg5_direct_pairs_counters.txt) shows an L2 hit rate of 13 %, a UTCL1 (L1 TLB)
// miss on 68 % of the accesses and 2.3e9 sector fetches per 1e9 points = 43 G random DRAM accesses per second, 34 % of
// the byte peak for 8 useful bytes in 64.  The same kernel runs 4.2x faster when the map fits an XCD's L2
// (profiles/r02_sampler_footprint.txt).  At config 5's density (1.07 points per pixel) the points can be brought to
// the map instead: a counting pass + a scatter pass put the points into 2-D map TILES of about an L2's worth, the
// gather then runs tile by tile out of L2 (or out of LDS), and a last pass returns the values to the caller's order.
// The per-point arithmetic is the direct kernel's, operation for operation, so the results are bit-identical to it
// (and to the oracle).  The tile index is only a locality hint: a point with any coordinates (outside the map, NaN)
// lands in SOME tile and is then sampled by the same general tap logic.
//
//   k_bin_count    sky -> (x, y) -> tile; per-chunk histogram in LDS -> cnt[chunk][tile] (u16)
//   k_bin_segsum / k_bin_scan / k_bin_offsets   column-wise exclusive scan -> off[chunk][tile] (u32)
//   k_bin_scatter  sky -> (x, y) -> tile; rank by LDS atomic; rec[off + rank] = (x, y); slot[k] = off + rank
//   k_sample_binned     rec (tile order) -> 2x2 taps out of the L2-resident tile -> val (tile order, coalesced)
//   k_sample_tile_regs  the same gather with records in registers and the tile's rows streamed through LDS (LDS-DMA)
//   k_bin_unpermute     out[k] = val[slot[k]]
// No global atomics anywhere; slots are a deterministic function of (chunk, tile) plus an arbitrary rank inside a
// run, which changes where a record sits, never its value.
//
// What was measured (1e9 points, 43200 x 21601 Float64 map, profiles/r02_cfg5_binned_*.txt): the TLB
// and DRAM problems do go away (UTCL1 misses 3.0e9 -> 1.6e4 in the gather, L2 hit rate 13 % -> 90 %), but every
// per-point DIVERGENT access costs about the same whatever it hits: a CU's L1 keeps only so many misses in flight, so
// 64 lanes x 64 different lines run at 110-150 G lane-accesses/s chip-wide out of L2 (70 G/s for 16-byte stores).  The
// binned pipeline pays six of those per point (1 record store, 4 taps, 1 value fetch) against the direct kernel's four:
// count 4.2 + tables 1.1 + scatter 15.7 + gather 18.5 + un-permute 11.7 = 51 ms against 53.7 ms direct.  Staging the
// taps in LDS instead (k_sample_tile_regs, bit-identical) trades the taps for re-reading the tile's rows once per
// 16 Ki records (67 GB of L2 -> LDS traffic): 26 ms, of which staging 9.6, strip tests + LDS taps 8.1, record load /
// value store 8.1, none of it overlapped at one workgroup per CU.  Kept as a plan the caller may choose; the
// default scattered entry stays the direct kernel.
#pragma once

struct BinGrid {
    int32_t tw, th;        // tile width / height in pixels
    int32_t TX, TY, B;     // tiles along RA / DEC, B = TX * TY
    float inv_tw, inv_th;
};

// tile of a 1-based pixel coordinate pair: a locality hint, clamped into the grid
__device__ inline int bin_of(const BinGrid& g, double x, double y) {
    int32_t i0, j0; double f;
    split_cell(x, &i0, &f);
    split_cell(y, &j0, &f);
    int bx = (int)((float)(i0 - 1) * g.inv_tw), by = (int)((float)(j0 - 1) * g.inv_th);
    bx = bx < 0 ? 0 : (bx >= g.TX ? g.TX - 1 : bx);
    by = by < 0 ? 0 : (by >= g.TY ? g.TY - 1 : by);
    return by * g.TX + bx;
}

#define PXL_BIN_THREADS 1024

// ---- pass 0: per-chunk tile histogram.  Chunk = PT * 1024 consecutive points of the batch.
template <int PT>
__global__ __launch_bounds__(PXL_BIN_THREADS) void k_bin_count(Sky2Pix s, BinGrid g, int64_t n,
                                                               const double2* __restrict__ sky,
                                                               uint16_t* __restrict__ cnt) {
    extern __shared__ uint32_t bin_lds[];
    uint32_t* lcnt = bin_lds;
    for (int b = threadIdx.x; b < g.B; b += PXL_BIN_THREADS) lcnt[b] = 0;
    __syncthreads();
    const int64_t chunk0 = (int64_t)blockIdx.x * (PT * PXL_BIN_THREADS);
#pragma unroll
    for (int j0 = 0; j0 < PT; j0 += 4) {
        double2 ad[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t k = chunk0 + (int64_t)(j0 + u) * PXL_BIN_THREADS + threadIdx.x;
            ad[u] = (k < n) ? sky[k] : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t k = chunk0 + (int64_t)(j0 + u) * PXL_BIN_THREADS + threadIdx.x;
            if (k < n) atomicAdd(&lcnt[bin_of(g, s2p_x(s, ad[u].x), s2p_y(s, ad[u].y))], 1u);
        }
    }
    __syncthreads();
    uint16_t* row = cnt + (int64_t)blockIdx.x * g.B;
    for (int b = threadIdx.x; b < g.B; b += PXL_BIN_THREADS) row[b] = (uint16_t)lcnt[b];
}

// ---- column-wise exclusive scan of cnt[W][B] over the chunks, chunk order = (chunk % 8, chunk / 8) when vmajor:
// runs of chunks w and w + 8 are then neighbours inside a tile's records.  Blocks b and b + 8 usually share an XCD
// (MI355X deals workgroups round-robin over its 8 XCDs), so the partially written 128-byte lines where two runs
// meet are completed inside ONE L2 rather than written back byte-masked by two.  Speed only, never correctness.
__device__ inline int64_t bin_row_of(int64_t p, int64_t W, int64_t Wv, int vmajor) {      // position -> chunk, -1 = none
    if (!vmajor) return p < W ? p : -1;
    const int64_t v = p / Wv, i = p - v * Wv, w = i * 8 + v;
    return w < W ? w : -1;
}
// Four adjacent tiles per thread (one 8-byte load of u16 counts per chunk row); B is padded to a multiple of 4 by the host.
__global__ __launch_bounds__(256) void k_bin_segsum(const uint16_t* __restrict__ cnt, int64_t W, int B, int64_t L,
                                                    int vmajor, uint32_t* __restrict__ seg) {
    const int b = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (b >= B) return;
    const int64_t Wv = (W + 7) / 8, NP = vmajor ? 8 * Wv : W;
    const int64_t p0 = (int64_t)blockIdx.y * L, p1 = (p0 + L < NP) ? p0 + L : NP;
    uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (int64_t p = p0; p < p1; ++p) {
        const int64_t w = bin_row_of(p, W, Wv, vmajor);
        if (w < 0) continue;
        const ushort4 c = *reinterpret_cast<const ushort4*>(cnt + w * B + b);
        s0 += c.x; s1 += c.y; s2 += c.z; s3 += c.w;
    }
    *reinterpret_cast<uint4*>(seg + (int64_t)blockIdx.y * B + b) = make_uint4(s0, s1, s2, s3);
}
// one block: per tile, exclusive scan over the S segments; then exclusive scan of the tile totals -> start[B + 1]
// (the totals are parked in start[] between the two steps)
__global__ __launch_bounds__(1024) void k_bin_scan(uint32_t* __restrict__ seg, int S, int B, uint32_t* __restrict__ start) {
    __shared__ uint32_t part[1024];
    for (int b = threadIdx.x; b < B; b += 1024) {
        uint32_t run = 0;
        for (int sgm = 0; sgm < S; ++sgm) { const uint32_t t = seg[(int64_t)sgm * B + b]; seg[(int64_t)sgm * B + b] = run; run += t; }
        start[b] = run;
    }
    __syncthreads();
    const int per = (B + 1023) / 1024;                 // consecutive tiles per thread
    uint32_t mine = 0;
    for (int q = 0; q < per; ++q) { const int b = threadIdx.x * per + q; if (b < B) mine += start[b]; }
    part[threadIdx.x] = mine;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {               // Hillis-Steele inclusive scan of the 1024 partials
        const uint32_t v = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - mine;
    for (int q = 0; q < per; ++q) {
        const int b = threadIdx.x * per + q;
        if (b < B) { const uint32_t t = start[b]; start[b] = run; run += t; }
    }
    if (threadIdx.x == 1023) start[B] = part[1023];
}
__global__ __launch_bounds__(256) void k_bin_offsets(const uint16_t* __restrict__ cnt, int64_t W, int B, int64_t L,
                                                     int vmajor, const uint32_t* __restrict__ seg,
                                                     const uint32_t* __restrict__ start, uint32_t* __restrict__ off) {
    const int b = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (b >= B) return;
    const int64_t Wv = (W + 7) / 8, NP = vmajor ? 8 * Wv : W;
    const int64_t p0 = (int64_t)blockIdx.y * L, p1 = (p0 + L < NP) ? p0 + L : NP;
    const uint4 sg = *reinterpret_cast<const uint4*>(seg + (int64_t)blockIdx.y * B + b);
    uint32_t r0 = start[b] + sg.x, r1 = start[b + 1] + sg.y, r2 = start[b + 2] + sg.z, r3 = start[b + 3] + sg.w;
    for (int64_t p = p0; p < p1; ++p) {
        const int64_t w = bin_row_of(p, W, Wv, vmajor);
        if (w < 0) continue;
        const ushort4 c = *reinterpret_cast<const ushort4*>(cnt + w * B + b);
        *reinterpret_cast<uint4*>(off + w * B + b) = make_uint4(r0, r1, r2, r3);
        r0 += c.x; r1 += c.y; r2 += c.z; r3 += c.w;
    }
}

// ---- pass 1: scatter.  off[chunk][tile] is known before the sweep, so a point's slot (off + its rank from an LDS
// atomic) is known as soon as its tile is: one sweep, nothing held in registers.
template <int PT>
__global__ __launch_bounds__(PXL_BIN_THREADS) void k_bin_scatter(Sky2Pix s, BinGrid g, int64_t n,
                                                                 const double2* __restrict__ sky,
                                                                 const uint32_t* __restrict__ off,
                                                                 double2* __restrict__ rec, uint32_t* __restrict__ slot) {
    extern __shared__ uint32_t bin_lds[];
    uint32_t* cursor = bin_lds;               // next free slot of this chunk's run in every tile
    const uint32_t* row = off + (int64_t)blockIdx.x * g.B;
    for (int b = threadIdx.x; b < g.B; b += PXL_BIN_THREADS) cursor[b] = row[b];
    __syncthreads();
    const int64_t chunk0 = (int64_t)blockIdx.x * (PT * PXL_BIN_THREADS);
#pragma unroll 1
    for (int j0 = 0; j0 < PT; j0 += 4) {
        double2 ad[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t k = chunk0 + (int64_t)(j0 + u) * PXL_BIN_THREADS + threadIdx.x;
            ad[u] = (k < n) ? sky[k] : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t k = chunk0 + (int64_t)(j0 + u) * PXL_BIN_THREADS + threadIdx.x;
            if (k < n) {
                const double x = s2p_x(s, ad[u].x), y = s2p_y(s, ad[u].y);
                const uint32_t sl = atomicAdd(&cursor[bin_of(g, x, y)], 1u);
                rec[sl] = make_double2(x, y);
                slot[k] = sl;
            }
        }
    }
}

// ---- pass 2: the gather, in tile order.  The arithmetic is k_sample_bilinear's from (x, y) on.
// Chunk order is XCD-aware like the reprojection's: blocks b and b + 8 take neighbouring chunks, so each XCD walks
// one contiguous eighth of the records, tile after tile, and its L2 holds its own current tile only.
#define PXL_BIN_SUNR 4
template <typename T>
__global__ __launch_bounds__(256) void k_sample_binned(const T* __restrict__ src, int64_t nx, int64_t ny, int32_t nc,
                                                       int64_t row0, int64_t nrows, int periodic, int64_t n,
                                                       const double2* __restrict__ rec, T* __restrict__ val,
                                                       int64_t nblk8, int trips) {
    const int64_t plane = nx * nrows;
    const int64_t cid = (int64_t)(blockIdx.x & 7) * nblk8 + (blockIdx.x >> 3);
    const int64_t chunk = (int64_t)blockDim.x * PXL_BIN_SUNR;
    for (int t = 0; t < trips; ++t) {
        const int64_t k0 = (cid * trips + t) * chunk + threadIdx.x;
        if (k0 - threadIdx.x >= n) break;
        double2 xy[PXL_BIN_SUNR];
#pragma unroll
        for (int u = 0; u < PXL_BIN_SUNR; ++u) {
            const int64_t k = k0 + u * blockDim.x;
            xy[u] = (k < n) ? rec[k] : make_double2(1.0, 1.0);
        }
        int64_t o00[PXL_BIN_SUNR], o10[PXL_BIN_SUNR], o01[PXL_BIN_SUNR], o11[PXL_BIN_SUNR];
        double fx[PXL_BIN_SUNR], fy[PXL_BIN_SUNR];
        bool fin[PXL_BIN_SUNR];
#pragma unroll
        for (int u = 0; u < PXL_BIN_SUNR; ++u) {
            const double x = xy[u].x, y = xy[u].y;
            fin[u] = isfinite(x) && isfinite(y);
            int32_t i0, j0;
            split_cell(x, &i0, &fx[u]);
            split_cell(y, &j0, &fy[u]);
            int64_t ia = i0, ib = (int64_t)i0 + 1;
            bool oka = true, okb = true;
            if (periodic) { ia = wrap_col(ia, nx); ib = wrap_col(ib, nx); }
            else { oka = (ia >= 1 && ia <= nx); okb = (ib >= 1 && ib <= nx); }
            const int64_t ja = (int64_t)j0 - 1 - row0, jb = ja + 1;
            const bool rowa = (j0 >= 1 && j0 <= ny && ja >= 0 && ja < nrows);
            const bool rowb = ((int64_t)j0 + 1 >= 1 && (int64_t)j0 + 1 <= ny && jb >= 0 && jb < nrows);
            o00[u] = (rowa && oka) ? ja * nx + (ia - 1) : -1;
            o10[u] = (rowa && okb) ? ja * nx + (ib - 1) : -1;
            o01[u] = (rowb && oka) ? jb * nx + (ia - 1) : -1;
            o11[u] = (rowb && okb) ? jb * nx + (ib - 1) : -1;
        }
        for (int c = 0; c < nc; ++c) {
            const T* pl = src + (int64_t)c * plane;
            double m00[PXL_BIN_SUNR], m10[PXL_BIN_SUNR], m01[PXL_BIN_SUNR], m11[PXL_BIN_SUNR];
#pragma unroll
            for (int u = 0; u < PXL_BIN_SUNR; ++u) {
                m00[u] = o00[u] >= 0 ? (double)pl[o00[u]] : 0.0;
                m10[u] = o10[u] >= 0 ? (double)pl[o10[u]] : 0.0;
                m01[u] = o01[u] >= 0 ? (double)pl[o01[u]] : 0.0;
                m11[u] = o11[u] >= 0 ? (double)pl[o11[u]] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < PXL_BIN_SUNR; ++u) {
                const int64_t k = k0 + u * blockDim.x;
                const double top = (1 - fx[u]) * m00[u] + fx[u] * m10[u];
                const double bot = (1 - fx[u]) * m01[u] + fx[u] * m11[u];
                const double v = (1 - fy[u]) * top + fy[u] * bot;
                if (k < n) val[(int64_t)c * n + k] = (T)(fin[u] ? v : __builtin_nan(""));
            }
        }
    }
}

// ---- general single-point path (the oracle's tap logic against global memory), used by the LDS form below for
// records whose cell is not inside the tile they were binned into (points outside the map, non-finite ones, float
// rounding of the tile index at a tile edge).
template <typename T>
__device__ inline double sample_point_global(const T* __restrict__ pl, int64_t nx, int64_t ny, int64_t row0, int64_t nrows,
                                             int periodic, int32_t i0, double fx, int32_t j0, double fy) {
    int64_t ia = i0, ib = (int64_t)i0 + 1;
    bool oka = true, okb = true;
    if (periodic) { ia = wrap_col(ia, nx); ib = wrap_col(ib, nx); }
    else { oka = (ia >= 1 && ia <= nx); okb = (ib >= 1 && ib <= nx); }
    const int64_t ja = (int64_t)j0 - 1 - row0, jb = ja + 1;
    const bool rowa = (j0 >= 1 && j0 <= ny && ja >= 0 && ja < nrows);
    const bool rowb = ((int64_t)j0 + 1 >= 1 && (int64_t)j0 + 1 <= ny && jb >= 0 && jb < nrows);
    const double m00 = (rowa && oka) ? (double)pl[ja * nx + (ia - 1)] : 0.0;
    const double m10 = (rowa && okb) ? (double)pl[ja * nx + (ib - 1)] : 0.0;
    const double m01 = (rowb && oka) ? (double)pl[jb * nx + (ia - 1)] : 0.0;
    const double m11 = (rowb && okb) ? (double)pl[jb * nx + (ib - 1)] : 0.0;
    const double top = (1 - fx) * m00 + fx * m10;
    const double bot = (1 - fx) * m01 + fx * m11;
    return (1 - fy) * top + fy * bot;
}

// one record through the general path, all components (rare: points outside the map, non-finite ones, tile-edge
// rounding); out of line so that the unrolled callers stay small
template <typename T>
__device__ __noinline__ void sample_record_general(const T* __restrict__ src, int64_t nx, int64_t ny, int32_t nc, int64_t row0,
                                                   int64_t nrows, int periodic, int64_t n, double x, double y, int64_t q,
                                                   T* __restrict__ val) {
    int32_t i0, j0; double gx, gy;
    split_cell(x, &i0, &gx);
    split_cell(y, &j0, &gy);
    const bool fin = isfinite(x) && isfinite(y);
    for (int c = 0; c < nc; ++c) {
        const double vv = sample_point_global(src + (int64_t)c * nx * nrows, nx, ny, row0, nrows, periodic, i0, gx, j0, gy);
        val[(int64_t)c * n + q] = (T)(fin ? vv : __builtin_nan(""));
    }
}

// ---- pass 3: back to the caller's order.  slot[k] of neighbouring k inside one chunk point into that chunk's runs,
// so the 8-byte gathers hit lines the same workgroup has just touched (L1 / L2), not DRAM.
template <typename T>
__global__ __launch_bounds__(256) void k_bin_unpermute(int64_t n, int32_t nc, const uint32_t* __restrict__ slot,
                                                       const T* __restrict__ val, T* __restrict__ out) {
    const int64_t chunk = (int64_t)blockDim.x * 4;
    for (int64_t k0 = (int64_t)blockIdx.x * chunk + threadIdx.x; k0 < n; k0 += (int64_t)gridDim.x * chunk) {
        uint32_t sl[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int64_t k = k0 + u * blockDim.x; sl[u] = (k < n) ? slot[k] : 0u; }
        for (int c = 0; c < nc; ++c) {
            T v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = val[(int64_t)c * n + sl[u]];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int64_t k = k0 + u * blockDim.x; if (k < n) out[(int64_t)c * n + k] = v[u]; }
        }
    }
}

// ---- work items of the register-resident gather: tile t contributes ceil(count_t / R) items; wstart = exclusive scan
__global__ __launch_bounds__(1024) void k_bin_items_scan(const uint32_t* __restrict__ start, int B, int R,
                                                         uint32_t* __restrict__ wstart) {
    __shared__ uint32_t part[1024];
    const int per = (B + 1023) / 1024;
    uint32_t mine = 0;
    for (int q = 0; q < per; ++q) {
        const int b = threadIdx.x * per + q;
        if (b < B) mine += (start[b + 1] - start[b] + R - 1) / R;
    }
    part[threadIdx.x] = mine;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const uint32_t v = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - mine;
    for (int q = 0; q < per; ++q) {
        const int b = threadIdx.x * per + q;
        if (b < B) { wstart[b] = run; run += (start[b + 1] - start[b] + R - 1) / R; }
    }
    if (threadIdx.x == 1023) wstart[B] = part[1023];
}
__global__ __launch_bounds__(256) void k_bin_items_fill(const uint32_t* __restrict__ wstart, int B, uint32_t* __restrict__ item) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    for (uint32_t i = wstart[b]; i < wstart[b + 1]; ++i) item[i] = (uint32_t)b;
}

// ---- pass 2, register form: a workgroup keeps RT records per thread of ONE tile in registers (cell and fractions
// computed once) and streams the tile's source rows through LDS strip by strip; a record takes its four taps out of
// LDS in the strip that holds its rows.  The records are read once (from HBM); the re-read traffic is the tile's source rows -- tile bytes per RT * 1024 records, coalesced, L2-resident because
// the workgroups of one tile are neighbours on one XCD -- and only the cheap strip test is repeated per strip.
// 512 threads = 2 waves per SIMD: 256 VGPRs per thread, room for RT = 32 records (5 VGPRs each) without spilling
#define PXL_TREG_THREADS 512
template <typename T, int RT, bool ONEC, bool DMA>
__global__ __launch_bounds__(PXL_TREG_THREADS) void k_sample_tile_regs(const T* __restrict__ src, int64_t nx, int64_t ny,
                                                                       int32_t nc, int64_t row0, int64_t nrows, int periodic,
                                                                       int64_t n, BinGrid g, int SH, int NS, int pitch,
                                                                       const uint32_t* __restrict__ start,
                                                                       const uint32_t* __restrict__ wstart,
                                                                       const uint32_t* __restrict__ item, uint32_t nitems8,
                                                                       const double2* __restrict__ rec, T* __restrict__ val,
                                                                       const void* __restrict__ zero_page, int dbg) {
    extern __shared__ __attribute__((aligned(16))) double strip_lds_raw[];
    T* L = reinterpret_cast<T*>(strip_lds_raw);
    // XCD-contiguous item order (blocks b and b + 8 share an XCD): one XCD works through neighbouring items = one tile
    const uint32_t it = (blockIdx.x & 7) * nitems8 + (blockIdx.x >> 3);
    if (it >= wstart[g.B]) return;
    const int t = (int)item[it];
    const int64_t R = (int64_t)RT * PXL_TREG_THREADS;
    const int64_t qa = (int64_t)start[t] + (int64_t)(it - wstart[t]) * R;
    const int64_t qe = (int64_t)start[t + 1];
    const int bx = t % g.TX, by = t / g.TX;
    const int64_t c0 = (int64_t)bx * g.tw, rtile = (int64_t)by * g.th;
    const int64_t plane = nx * nrows;
    // records -> (packed cell inside the tile, fractions); records not inside this tile take the general path at once
    double fx[RT], fy[RT];
    int32_t pk[RT];
#pragma unroll
    for (int u0 = 0; u0 < RT; u0 += 4) {
        double2 xy[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t q = qa + (int64_t)(u0 + u) * PXL_TREG_THREADS + threadIdx.x;
            xy[u] = (q < qe) ? rec[q] : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t q = qa + (int64_t)(u0 + u) * PXL_TREG_THREADS + threadIdx.x;
            int32_t i0, j0;
            split_cell(xy[u].x, &i0, &fx[u0 + u]);
            split_cell(xy[u].y, &j0, &fy[u0 + u]);
            const int64_t ci = (int64_t)i0 - 1 - c0, rt = (int64_t)j0 - 1 - rtile;
            const bool in_tile = (ci >= 0 && ci < g.tw && rt >= 0 && rt < g.th) && isfinite(xy[u].x) && isfinite(xy[u].y);
            int32_t code = in_tile ? (int32_t)((ci << 12) | rt) : -2;          // th <= 4096, tw < 2^19 (host-checked)
            if (q < qe && !in_tile) sample_record_general(src, nx, ny, nc, row0, nrows, periodic, n, xy[u].x, xy[u].y, q, val);
            pk[u0 + u] = (q < qe) ? code : -2;                                   // -2: nothing (left) to do
        }
    }
    // ONEC (one component): a record's value replaces its fx once computed (pk = -3) and all values leave in one
    // coalesced store at the end; with several components the values are stored as they are computed.
    for (int c = 0; c < nc; ++c) {
        const T* pl = src + (int64_t)c * plane;
        for (int sidx = 0; sidx < NS; ++sidx) {
            const int64_t rbase = rtile + (int64_t)sidx * SH;
            __syncthreads();                                   // the previous strip has been consumed
            if (dbg & 1) {
            } else if (DMA) {
                // LDS-DMA (global_load_lds_dwordx4): every wave instruction moves 1 KiB of one source row straight
                // into the strip, no VGPRs, and a wave keeps all its instructions of the strip in flight at once.
                // Requires 16-byte alignment of every row segment: nx and tw multiples of EPL, src 16-byte aligned
                // (host-checked).  A lane whose pair lies outside the map or the resident window reads the zero page;
                // on a periodic map columns past nx continue at column 1 (pairs never straddle the seam).
                constexpr int EPL = 16 / (int)sizeof(T);
                const int nch = pitch / (64 * EPL);                    // 1-KiB pieces per strip row
                const int npiece = (SH + 1) * nch;
                const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
                const uint32_t lds0 = (uint32_t)(uintptr_t)L;          // LDS byte address of the strip
                for (int pc = wave; pc < npiece; pc += PXL_TREG_THREADS / 64) {
                    const int r = pc / nch, k = pc - r * nch;
                    const int64_t jj = rbase + 1 + r, jr = jj - 1 - row0;
                    const bool rowok = (jj >= 1 && jj <= ny && jr >= 0 && jr < nrows);
                    int64_t col = c0 + (int64_t)k * (64 * EPL) + (int64_t)EPL * lane;      // 0-based column of the lane's first element
                    bool ok = rowok;
                    if (col >= nx) { if (periodic && col < 2 * nx) col -= nx; else ok = false; }
                    const void* gsrc = ok ? (const void*)(pl + jr * nx + col) : zero_page;
                    glds16_vaddr(gsrc, (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds0 + (uint32_t)((r * pitch + k * 64 * EPL) * (int)sizeof(T)))));
                }
                wait_vm<0>();
            } else {
                const int wcols = g.tw + 1, nelem = (SH + 1) * wcols;
                const float inv_w = 1.0f / (float)wcols;
                for (int e0 = 0; e0 < nelem; e0 += PXL_TREG_THREADS * 8) {
                    T vals[8];
                    int dst[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int e = e0 + u * PXL_TREG_THREADS + threadIdx.x;
                        int r = (int)((float)e * inv_w);
                        int cc = e - r * wcols;
                        if (cc < 0) { --r; cc += wcols; } else if (cc >= wcols) { ++r; cc -= wcols; }
                        const int64_t jj = rbase + 1 + r, jr = jj - 1 - row0;
                        bool ok = e < nelem && (jj >= 1 && jj <= ny && jr >= 0 && jr < nrows);
                        int64_t i = c0 + 1 + cc;
                        if (periodic) i = wrap_col(i, nx);
                        else ok = ok && (i >= 1 && i <= nx);
                        vals[u] = ok ? pl[jr * nx + (i - 1)] : (T)0;
                        dst[u] = e < nelem ? r * pitch + cc : -1;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) if (dst[u] >= 0) L[dst[u]] = vals[u];
                }
            }
            __syncthreads();
            if (dbg & 2) continue;
#pragma unroll
            for (int u = 0; u < RT; ++u) {
                const int rj = (pk[u] & 0xfff) - sidx * SH;
                if (pk[u] >= 0 && rj >= 0 && rj < SH) {
                    const T* a = L + rj * pitch + (pk[u] >> 12);
                    const double m00 = (double)a[0], m10 = (double)a[1], m01 = (double)a[pitch], m11 = (double)a[pitch + 1];
                    const double top = (1 - fx[u]) * m00 + fx[u] * m10;
                    const double bot = (1 - fx[u]) * m01 + fx[u] * m11;
                    const double vv = (1 - fy[u]) * top + fy[u] * bot;
                    if (ONEC) { fx[u] = vv; pk[u] = -3; }
                    else val[(int64_t)c * n + qa + (int64_t)u * PXL_TREG_THREADS + threadIdx.x] = (T)vv;
                }
                // keep the compiler from hoisting every record's LDS reads to the top of the unrolled loop (4 live
                // doubles per record: hundreds of spilled registers)
                if ((u & 3) == 3) asm volatile("" ::: "memory");
            }
        }
#pragma unroll
        for (int u = 0; u < RT; ++u) {
            const int64_t q = qa + (int64_t)u * PXL_TREG_THREADS + threadIdx.x;
            if (pk[u] == -3) val[(int64_t)c * n + q] = (T)fx[u];
        }
    }
}
