// pxl_reproject.h -- separable CAR->CAR reprojection: tables, parameters, direct-gather and register-staged kernels; included by pxl_kernels.hip (one translation unit, -ffp-contract=off).
#pragma once

// ------------------------------------------------------------------------------------------------
// Reprojection (R1).
// ------------------------------------------------------------------------------------------------
// Separable tables: for output column i (0-based ic) the source cell xi0[ic] (1-based int) and fraction
// xfx[ic]; same for rows.  (a, d) = pix2sky(out; safe=false) [car_proj.jl:146-147];
// (x, y) = sky2pix(in; safe=true), division form [car_proj.jl:225-231].
__global__ __launch_bounds__(256) void k_build_tables(CarAffine out, Sky2Pix in, int64_t nxo, int64_t nyo,
                                                      int32_t* __restrict__ xi0, double* __restrict__ xfx,
                                                      int32_t* __restrict__ yj0, double* __restrict__ yfy) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nxo + nyo; k += stride) {
        if (k < nxo) {
            double a = p2s_ra(out, (double)(k + 1));
            split_cell(s2p_x(in, a), &xi0[k], &xfx[k]);
        } else {
            int64_t j = k - nxo;
            double d = p2s_dec(out, (double)(j + 1));
            split_cell(s2p_y(in, d), &yj0[j], &yfy[j]);
        }
    }
}

struct ReprojParams {
    const void* src;       // (nx, src_nrows, nc), Float64 or Float32 storage
    void* dst;             // (nxo, dst_nrows, nc), same storage type
    const int32_t* xi0; const double* xfx;   // nxo entries
    const int32_t* yj0; const double* yfy;   // nyo entries (absolute output row)
    int64_t nx, ny, src_row0, src_nrows;
    int64_t nxo, dst_row0, dst_nrows;
    int64_t r0, nr;        // output rows handled by this launch, relative to the dst window
    int32_t nc, periodic;
    // staged kernel only
    int32_t rh;            // output rows per tile
    int32_t seg;           // LDS slot length in doubles (even)
    int32_t dxpos;         // source column increases with output column
    int32_t dypos;         // source row increases with output row
    int32_t ntx, nty;      // tiles along RA / DEC
    int64_t ntiles, tiles_per_xcd;
    int64_t xchunk;        // tiles an XCD takes in one piece (xcd_tile below); tiles_per_xcd = one contiguous eighth each
    int32_t flags;         // tuning/diagnostics: 1 = skip source loads, 2 = skip stores, 4 = no XCD remap, 8 / 16 = tile order 1 / 2 of xcd_tile, 64 = stores only
    // LDS-DMA kernel only
    int32_t ns, pf;        // ring slots (power of two), prefetch distance in output rows
    int32_t nt;            // non-temporal stores in full tiles
    const double* zero_page;   // 16 bytes of zeros in device memory
};

// Block -> tile with XCD affinity.  Blocks b and b + 8 share an XCD (workgroups are dealt round-robin over the 8 XCDs),
// so XCD v = b % 8 is given the tiles of chunks v, v + 8, v + 16, ... of `chunk` consecutive tiles each: inside a
// chunk neighbouring tiles (shared 128-byte lines, shared halo rows) meet in one L2.  chunk = ntiles / 8 is one
// contiguous eighth of the map per XCD (round 1); smaller chunks keep the eight write fronts a chunk apart instead of
// an eighth of the map apart.  The grid must cover ceil(ntiles / (8 chunk)) * 8 chunk blocks.
// order: 0 = every XCD walks its piece upwards from its start (eight write fronts a piece apart, moving in lockstep);
// 1 = odd XCDs walk downwards (the distances between fronts change all the time); 2 = XCD v starts v/8 of the way into
// its piece and wraps around (fronts 9/8 of a piece apart).  Experiments on the write-placement effect (history: docs/DESIGN_history_r01-r03.md 9 item 6).
__device__ inline int64_t xcd_tile(int64_t b, int64_t chunk, int order = 0) {
    const int64_t v = b & 7, j = b >> 3;
    const int64_t c = j / chunk;
    int64_t w = j - c * chunk;
    if (order == 1 && (v & 1)) w = chunk - 1 - w;
    else if (order == 2) { w += v * (chunk / 8); if (w >= chunk) w -= chunk; }
    return (c * 8 + v) * chunk + w;
}

// ---- generic direct-gather kernel: one lane per output pixel pair, 4 taps from global memory each.
//      Used when a tile's source footprint does not fit the LDS ring (large down-scaling) and as the
//      cross-check variant.
template <typename T>
__global__ __launch_bounds__(256) void k_reproject_gather(ReprojParams p) {
    const int64_t npair = (p.nxo + 1) / 2;
    const int64_t total = npair * p.nr;
    const int c = blockIdx.y;
    SrcViewT<T> m{(const T*)p.src + (int64_t)c * p.nx * p.src_nrows, p.nx, p.ny, p.src_row0, p.src_nrows, p.periodic};
    T* dplane = (T*)p.dst + (int64_t)c * p.nxo * p.dst_nrows;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        int64_t rr = t / npair;
        int64_t i = (t - rr * npair) * 2;
        int64_t r = p.r0 + rr;
        int64_t j0 = p.yj0[p.dst_row0 + r];
        double fy = p.yfy[p.dst_row0 + r];
        int64_t o = r * p.nxo + i;
        dplane[o] = (T)bilerp_cells(m, p.xi0[i], p.xfx[i], j0, fy);
        if (i + 1 < p.nxo) dplane[o + 1] = (T)bilerp_cells(m, p.xi0[i + 1], p.xfx[i + 1], j0, fy);
    }
}

// ---- staged kernel: ONE WAVEFRONT PER OUTPUT TILE.
//
// A tile is TW = 128*PAIRS output columns x rh output rows of one component plane.  The wave marches
// down the tile's rows.  The two source rows an output row needs (j0, j0+1) live in a 4-slot LDS ring
// (slot = row & 3, tagged with the row id), each slot holding the contiguous source-column segment the
// tile's columns touch, loaded with 16 B/lane coalesced reads; the RA seam of a full-sky map is
// resolved while staging (segment column u -> u mod nx), so the interpolation itself never sees it.
// Rows for output row r+1 are prefetched into registers while row r is computed and stored.
// Output is written with 16 B/lane coalesced stores (lane = 2 adjacent RA pixels per PAIR).
//
// Tiles whose columns do not fit the slot (the rewind discontinuity of a partial-sky source falling
// inside the tile) fall back, wave-uniformly, to direct taps.
#define PXL_NS 4
#define PXL_MAXCH 5     // 16-B chunks of 64 lanes per slot: slot <= 5*128 doubles

template <bool VEC>
__device__ inline void load_row_regs(const ReprojParams& p, const double* plane, int64_t j, int64_t cbase0,
                                     int lane, double2 (&regs)[PXL_MAXCH]) {
    // j: 1-based absolute source row (any integer).  Rows outside the map / resident window read as 0.
    int64_t jr = j - 1 - p.src_row0;
    const bool row_ok = (j >= 1) && (j <= p.ny) && (jr >= 0) && (jr < p.src_nrows) && !(p.flags & 1);
    const double* rowp = plane + (row_ok ? jr : 0) * p.nx;
#pragma unroll
    for (int ch = 0; ch < PXL_MAXCH; ++ch) {
        int k = ch * 128 + 2 * lane;
        double2 v = make_double2(0.0, 0.0);
        if (k < p.seg && row_ok) {
            int64_t u = cbase0 + k;                   // 0-based unwrapped column of the chunk's first element
            if (VEC) {
                // nx even and u even: the pair never straddles the seam or the map edge
                bool ok = true;
                if (p.periodic) { u %= p.nx; if (u < 0) u += p.nx; }
                else ok = (u >= 0) && (u < p.nx);
                if (ok) v = *reinterpret_cast<const double2*>(rowp + u);
            } else {
                int64_t u0 = u, u1 = u + 1;
                bool ok0 = true, ok1 = true;
                if (p.periodic) {
                    u0 %= p.nx; if (u0 < 0) u0 += p.nx;
                    u1 %= p.nx; if (u1 < 0) u1 += p.nx;
                } else {
                    ok0 = (u0 >= 0) && (u0 < p.nx);
                    ok1 = (u1 >= 0) && (u1 < p.nx);
                }
                if (ok0) v.x = rowp[u0];
                if (ok1) v.y = rowp[u1];
            }
        }
        regs[ch] = v;
    }
}

__device__ inline void store_row_lds(const ReprojParams& p, double* slot, int lane, const double2 (&regs)[PXL_MAXCH]) {
#pragma unroll
    for (int ch = 0; ch < PXL_MAXCH; ++ch) {
        int k = ch * 128 + 2 * lane;
        if (k < p.seg) *reinterpret_cast<double2*>(slot + k) = regs[ch];
    }
}

template <int PAIRS, bool VEC>
__global__ __launch_bounds__(64) void k_reproject_staged(ReprojParams p) {
    extern __shared__ __attribute__((aligned(16))) double lds[];   // PXL_NS * seg doubles
    const int lane = threadIdx.x;
    constexpr int TW = 128 * PAIRS;

    // XCD-aware decode: hardware deals blocks round-robin over the 8 XCDs (b % 8); give each XCD a
    // contiguous run of tiles so RA-neighbouring tiles (which share 128-B lines at their edges and the
    // same source rows) hit the same L2.  Placement only affects speed, never correctness.
    const int64_t b = blockIdx.x;
    const int64_t t = (p.flags & 4) ? b : xcd_tile(b, p.xchunk, (p.flags & 8) ? 1 : ((p.flags & 16) ? 2 : 0));
    if (t >= p.ntiles) return;
    const int tx = (int)(t % p.ntx);
    const int64_t trest = t / p.ntx;
    const int ty = (int)(trest % p.nty);
    const int c = (int)(trest / p.nty);

    const double* splane = (const double*)p.src + (int64_t)c * p.nx * p.src_nrows;
    double* dplane = (double*)p.dst + (int64_t)c * p.nxo * p.dst_nrows;

    const int64_t c0 = (int64_t)tx * TW;                       // first output column of the tile
    const int64_t clast = (c0 + TW < p.nxo ? c0 + TW : p.nxo) - 1;
    const int64_t rb = p.r0 + (int64_t)ty * p.rh;              // rows relative to the dst window
    const int64_t re = (rb + p.rh < p.r0 + p.nr) ? rb + p.rh : p.r0 + p.nr;

    // ---- per-lane column setup
    const int64_t a = p.dxpos ? p.xi0[c0] : p.xi0[clast];      // 1-based source cell of the tile's low end
    const int64_t ua = a - 1;
    const int64_t cbase0 = ua & ~(int64_t)1;                   // even 0-based column at slot index 0
    int dloc[PAIRS][2];
    double fx[PAIRS][2];
    bool act[PAIRS][2];
    bool fits = true;
#pragma unroll
    for (int q = 0; q < PAIRS; ++q) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            int64_t col = c0 + q * 128 + 2 * lane + e;
            act[q][e] = col < p.nxo;
            int64_t i0 = act[q][e] ? p.xi0[col] : a;
            fx[q][e] = act[q][e] ? p.xfx[col] : 0.0;
            int64_t d = i0 - a;
            if (p.periodic) { d %= p.nx; if (d < 0) d += p.nx; }
            d += ua - cbase0;
            if (d < 0 || d + 1 >= p.seg) fits = false;
            dloc[q][e] = (int)d;
        }
    }
    const bool vec_store = ((p.nxo & 1) == 0) && (((uintptr_t)p.dst & 15) == 0);

    if (!__all(fits)) {
        // wave-uniform fallback: direct taps for this tile
        SrcView m{splane, p.nx, p.ny, p.src_row0, p.src_nrows, p.periodic};
        for (int64_t r = rb; r < re; ++r) {
            int64_t j0 = p.yj0[p.dst_row0 + r];
            double fy = p.yfy[p.dst_row0 + r];
#pragma unroll
            for (int q = 0; q < PAIRS; ++q)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    int64_t col = c0 + q * 128 + 2 * lane + e;
                    if (act[q][e]) dplane[r * p.nxo + col] = bilerp_cells(m, p.xi0[col], fx[q][e], j0, fy);
                }
        }
        return;
    }

    // ---- ring state (wave-uniform): tag of the source row held by each slot
    int64_t tag0 = INT64_MIN, tag1 = INT64_MIN, tag2 = INT64_MIN, tag3 = INT64_MIN;
    auto resident = [&](int64_t j) -> bool {
        int s = (int)(j & 3);
        int64_t tg = (s == 0) ? tag0 : (s == 1) ? tag1 : (s == 2) ? tag2 : tag3;
        return tg == j;
    };
    auto settag = [&](int64_t j) {
        int s = (int)(j & 3);
        if (s == 0) tag0 = j; else if (s == 1) tag1 = j; else if (s == 2) tag2 = j; else tag3 = j;
    };

    // per-row table entries of this tile live in lane (r - rb) and are broadcast with v_readlane
    // (no scalar-memory round trip inside the row loop); rh <= 64
    int my_j0 = 0;
    double my_fy = 0.0;
    if (rb + lane < re) { my_j0 = p.yj0[p.dst_row0 + rb + lane]; my_fy = p.yfy[p.dst_row0 + rb + lane]; }
    auto row_j0 = [&](int64_t r) -> int64_t { return (int64_t)__builtin_amdgcn_readlane(my_j0, (int)(r - rb)); };
    auto row_fy = [&](int64_t r) -> double {
        int lo = __builtin_amdgcn_readlane(__double2loint(my_fy), (int)(r - rb));
        int hi = __builtin_amdgcn_readlane(__double2hiint(my_fy), (int)(r - rb));
        return __hiloint2double(hi, lo);
    };

    double2 ra_[PXL_MAXCH], rb_[PXL_MAXCH];
    {   // prologue: rows of the first output row
        int64_t j0 = row_j0(rb);
        load_row_regs<VEC>(p, splane, j0, cbase0, lane, ra_);
        load_row_regs<VEC>(p, splane, j0 + 1, cbase0, lane, rb_);
        store_row_lds(p, lds + (j0 & 3) * p.seg, lane, ra_);
        store_row_lds(p, lds + ((j0 + 1) & 3) * p.seg, lane, rb_);
        settag(j0); settag(j0 + 1);
        __syncthreads();
    }

    for (int64_t r = rb; r < re; ++r) {
        const int64_t j0 = row_j0(r);
        const double fy = row_fy(r);

        // prefetch the rows output row r+1 needs and the ring lacks (global -> registers)
        bool needA = false, needB = false;
        int64_t jn = 0;
        if (r + 1 < re) {
            jn = row_j0(r + 1);
            needA = !resident(jn);
            needB = !resident(jn + 1);
            if (needA) load_row_regs<VEC>(p, splane, jn, cbase0, lane, ra_);
            if (needB) load_row_regs<VEC>(p, splane, jn + 1, cbase0, lane, rb_);
        }

        // interpolate output row r from LDS
        const double* T = lds + (j0 & 3) * p.seg;
        const double* B = lds + ((j0 + 1) & 3) * p.seg;
        const double wy = 1 - fy;
#pragma unroll
        for (int q = 0; q < PAIRS; ++q) {
            double v[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                int d = dloc[q][e];
                double wx = 1 - fx[q][e];
                double top = wx * T[d] + fx[q][e] * T[d + 1];
                double bot = wx * B[d] + fx[q][e] * B[d + 1];
                v[e] = wy * top + fy * bot;
            }
            int64_t col = c0 + q * 128 + 2 * lane;
            double* o = dplane + r * p.nxo + col;
            if (p.flags & 2) { if (v[0] == 1.2345e300) o[0] = v[1]; }       // diagnostics: keep v live, never store
            else if (vec_store) { if (act[q][0]) *reinterpret_cast<double2*>(o) = make_double2(v[0], v[1]); }
            else { if (act[q][0]) o[0] = v[0]; if (act[q][1]) o[1] = v[1]; }
        }

        if (needA || needB) {
            __syncthreads();                       // every lane is done reading the slots being replaced
            if (needA) { store_row_lds(p, lds + (jn & 3) * p.seg, lane, ra_); settag(jn); }
            if (needB) { store_row_lds(p, lds + ((jn + 1) & 3) * p.seg, lane, rb_); settag(jn + 1); }
            __syncthreads();
        }
    }
}
