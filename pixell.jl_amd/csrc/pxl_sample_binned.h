// pxl_sample_binned.h -- tile-binned scattered sampler (BASELINE config 5); included by pxl_kernels.hip (one
// translation unit, -ffp-contract=off).
//
// Why: the direct gather (k_sample_bilinear) touches ~2.1 random 64-byte DRAM sectors per point on a multi-GB map;
// rocprofv3 (profiles/r02_cfg5_counters.txt) shows an L2 hit rate of 13 %, a UTCL1 (L1 TLB) miss on 68 % of the
// accesses and 2.3e9 sector fetches per 1e9 points, i.e. 43 G random DRAM accesses per second -- the part's
// row-activate rate, 34 % of the byte peak for 8 useful bytes in 64.  The same kernel runs 4.2x faster when the map
// fits an XCD's L2 (profiles/r02_sampler_footprint.txt).  At config 5's density (1.07 points per pixel) the points
// can be brought to the map instead: one counting pass + one scatter pass put the points into map TILES of about an
// L2's worth (2-D tiles, a few thousand of them), the gather then runs tile by tile out of L2, and a last pass
// returns the values to the caller's order.  Every pass but the gather is a coalesced stream; the per-point
// arithmetic is the direct kernel's, operation for operation, so the results are bit-identical to it (and to the
// oracle).  The tile index is only a locality hint: a point with any coordinates (outside the map, NaN) lands in
// SOME tile and is then sampled by the same general tap logic.
//
//   k_bin_count    sky -> (x, y) -> tile; per-chunk histogram in LDS -> cnt[chunk][tile] (u16)
//   k_bin_segsum / k_bin_scan / k_bin_offsets   column-wise exclusive scan -> off[chunk][tile] (u32)
//   k_bin_scatter  sky -> (x, y) -> tile; rank by LDS atomic; rec[off + rank] = (x, y); slot[k] = off + rank
//   k_sample_binned  rec (tile order) -> 2x2 taps (L2-resident tile) -> val (tile order, coalesced)
//   k_bin_unpermute  out[k] = val[slot[k]]
// No global atomics anywhere; slots are a deterministic function of (chunk, tile) plus an arbitrary rank inside a
// run, which changes where a record sits, never its value.
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
__global__ __launch_bounds__(256) void k_bin_segsum(const uint16_t* __restrict__ cnt, int64_t W, int B, int64_t L,
                                                    int vmajor, uint32_t* __restrict__ seg) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const int64_t Wv = (W + 7) / 8, NP = vmajor ? 8 * Wv : W;
    const int64_t p0 = (int64_t)blockIdx.y * L, p1 = (p0 + L < NP) ? p0 + L : NP;
    uint32_t sum = 0;
    for (int64_t p = p0; p < p1; ++p) {
        const int64_t w = bin_row_of(p, W, Wv, vmajor);
        if (w >= 0) sum += cnt[w * B + b];
    }
    seg[(int64_t)blockIdx.y * B + b] = sum;
}
// one block: per tile, exclusive scan over the S segments; then exclusive scan of the tile totals -> start[B + 1]
__global__ __launch_bounds__(1024) void k_bin_scan(uint32_t* __restrict__ seg, int S, int B, uint32_t* __restrict__ start) {
    __shared__ uint32_t part[1024];
    __shared__ uint32_t tot[8192];
    for (int b = threadIdx.x; b < B; b += 1024) {
        uint32_t run = 0;
        for (int sgm = 0; sgm < S; ++sgm) { const uint32_t t = seg[(int64_t)sgm * B + b]; seg[(int64_t)sgm * B + b] = run; run += t; }
        tot[b] = run;
    }
    __syncthreads();
    const int per = (B + 1023) / 1024;                 // consecutive tiles per thread
    uint32_t mine = 0;
    for (int q = 0; q < per; ++q) { const int b = threadIdx.x * per + q; if (b < B) mine += tot[b]; }
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
        if (b < B) { start[b] = run; run += tot[b]; }
    }
    if (threadIdx.x == 1023) start[B] = part[1023];
}
__global__ __launch_bounds__(256) void k_bin_offsets(const uint16_t* __restrict__ cnt, int64_t W, int B, int64_t L,
                                                     int vmajor, const uint32_t* __restrict__ seg,
                                                     const uint32_t* __restrict__ start, uint32_t* __restrict__ off) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const int64_t Wv = (W + 7) / 8, NP = vmajor ? 8 * Wv : W;
    const int64_t p0 = (int64_t)blockIdx.y * L, p1 = (p0 + L < NP) ? p0 + L : NP;
    uint32_t run = start[b] + seg[(int64_t)blockIdx.y * B + b];
    for (int64_t p = p0; p < p1; ++p) {
        const int64_t w = bin_row_of(p, W, Wv, vmajor);
        if (w < 0) continue;
        const uint32_t c = cnt[w * B + b];
        off[w * B + b] = run;
        run += c;
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
