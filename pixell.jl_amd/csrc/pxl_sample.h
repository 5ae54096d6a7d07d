// pxl_sample.h -- non-separable paths: CAR<->Gnomonic reprojection and the scattered bilinear sampler; included by pxl_kernels.hip (one translation unit, -ffp-contract=off).
#pragma once

// ---- generic (non-separable) bilinear reprojection between CAR and Gnomonic maps (N2).
// Per output pixel: (ra, dec) = pix2sky(out) [car_proj.jl:146-147 safe=false | tan_proj.jl:59-75];
// (x, y) = sky2pix(in) [car_proj.jl:225-231 safe=true | tan_proj.jl:44-57]; 2x2 direct taps + lerp.
// A sky point behind a Gnomonic source's tangent plane (cos c <= 0) is not on that map: it reads as 0.
// k_reproject_generic evaluates the coordinates per pixel (about ten FP64 libm calls: transcendental bound);
// k_reproject_generic_tiled interpolates them per tile with a checked error bound (HBM / gather bound).
// Tolerance-checked rather than bit-exact either way.
struct GenericParams {
    const double* src; double* dst;
    int64_t nx, ny, nxo, nyo;
    int32_t nc, periodic, proj_in, proj_out;
    CarAffine out_car; TanParams out_tan;
    Sky2Pix in_car; TanParams in_tan;
    unsigned int* exact_tiles;         // counts the tiles that took the exact path (may be null)
    unsigned int* exact_tiles_next;    // the slot the NEXT call will count in: zeroed by this call's exact launch (no memset per call)
};
// exact source coordinates of output pixel (i, j) (1-based, may lie outside the output map): the evaluators of the
// reference, per pixel.  *visible: the sky point is in front of a Gnomonic source's tangent plane.
__device__ inline void generic_coords(const GenericParams& p, double i, double j, double* x, double* y, bool* visible) {
    double ra, dec;
    if (p.proj_out == PXL_PROJ_TAN) tan_pix2sky(p.out_tan, i, j, &ra, &dec);
    else { ra = p2s_ra(p.out_car, i); dec = p2s_dec(p.out_car, j); }
    *visible = true;
    if (p.proj_in == PXL_PROJ_TAN) {
        tan_sky2pix(p.in_tan, ra, dec, x, y);
        *visible = (p.in_tan.sd0 * sin(dec) + cos(dec) * cos(ra - p.in_tan.a0) * p.in_tan.cd0) > 0.0;
    } else { *x = s2p_x(p.in_car, ra); *y = s2p_y(p.in_car, dec); }
}
__device__ inline void generic_store(const GenericParams& p, int64_t t, double x, double y, bool visible) {
    const int64_t total = p.nxo * p.nyo;
    const bool fin = isfinite(x) && isfinite(y);
    int32_t i0, j0; double fx, fy;
    split_cell(x, &i0, &fx);
    split_cell(y, &j0, &fy);
    // interior cell (all four taps inside the map, or wrapping once on a periodic one): two row offsets, no per-tap checks
    const bool jin = j0 >= 1 && j0 < p.ny;
    const bool iin = p.periodic ? (i0 >= 0 && i0 <= p.nx) : (i0 >= 1 && i0 < p.nx);
#ifdef PXL_GENERIC_PAIR_TAPS
    // the two taps of a row are adjacent in memory unless the cell straddles the seam: ONE 16-byte load per row (8-byte aligned)
    if (visible && fin && jin && i0 >= 1 && i0 < p.nx) {
        struct __attribute__((packed, aligned(8))) Pair { double a, b; };
        const int64_t o = (int64_t)(j0 - 1) * p.nx + (i0 - 1);
        for (int c = 0; c < p.nc; ++c) {
            const double* pl = p.src + (int64_t)c * p.nx * p.ny;
            const Pair tp = *reinterpret_cast<const Pair*>(pl + o), bt = *reinterpret_cast<const Pair*>(pl + o + p.nx);
            const double top = (1 - fx) * tp.a + fx * tp.b;
            const double bot = (1 - fx) * bt.a + fx * bt.b;
            p.dst[(int64_t)c * total + t] = (1 - fy) * top + fy * bot;
        }
        return;
    }
#endif
    if (visible && fin && jin && iin) {
        const int64_t ia = (i0 >= 1 ? i0 : p.nx) - 1, ib = (i0 < p.nx ? i0 + 1 : 1) - 1;       // 0-based columns of i0, i0 + 1
        const int64_t ra = (int64_t)(j0 - 1) * p.nx, rb = ra + p.nx;
        for (int c = 0; c < p.nc; ++c) {
            const double* pl = p.src + (int64_t)c * p.nx * p.ny;
            const double top = (1 - fx) * pl[ra + ia] + fx * pl[ra + ib];
            const double bot = (1 - fx) * pl[rb + ia] + fx * pl[rb + ib];
            p.dst[(int64_t)c * total + t] = (1 - fy) * top + fy * bot;
        }
        return;
    }
    for (int c = 0; c < p.nc; ++c) {
        SrcView m{p.src + (int64_t)c * p.nx * p.ny, p.nx, p.ny, 0, p.ny, p.periodic};
        double v = visible ? bilerp_cells(m, i0, fx, j0, fy) : 0.0;
        p.dst[(int64_t)c * total + t] = fin ? v : __builtin_nan("");
    }
}
__global__ __launch_bounds__(256) void k_reproject_generic(GenericParams p) {
    const int64_t total = p.nxo * p.nyo;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const int64_t jr = t / p.nxo, i = t - jr * p.nxo;
        double x, y; bool visible;
        generic_coords(p, (double)(i + 1), (double)(jr + 1), &x, &y, &visible);
        generic_store(p, t, x, y, visible);
    }
}

// ---- the same operator with the coordinate map interpolated per output tile (what python-pixell does for non-
// separable reprojections, and the only way off the transcendental roof: ~10 FP64 libm calls per pixel above).
// Tile = 128 x 32 output pixels.  The exact (x, y) of the reference's
// evaluators is computed on a 7 x 6 lattice of the tile (54 evaluations per 4096 pixels with the check points) and
// every pixel takes the tensor-product Lagrange interpolant of degree 6 x 5 (truncation error ~1e-13 pixel at
// 0.5 arcmin, below the rounding noise of the exact evaluation; 64-wide tiles with 6 x 6 nodes cost twice the lattice
// work per pixel: 1.95 vs 1.75 ms on the 16-patch mosaic, and 128-wide tiles with only 6 nodes across fail the check
// on a quarter of the tiles of a 34-degree patch).  The interpolant is CHECKED per tile against exact evaluations at
// twelve off-lattice points: if either
// coordinate is off by more than PXL_TILED_TOL pixel anywhere, or a lattice point is non-finite or behind the
// tangent plane (the rewind jump of a periodic source and the Gnomonic horizon land here), the whole tile takes the
// exact per-pixel path -- so the result differs from k_reproject_generic by at most PXL_TILED_TOL pixel in the
// sampling position, i.e. ~1e-10 x the local map gradient in value, inside the reference's own Gnomonic tolerance
// (test/test_geometry.jl:116-119: 1e-9).
// 1e-10 pixel: the per-pixel evaluation itself carries ~1e-11 pixel of libm rounding noise at x ~ 2e4 (ulp 3.6e-12,
// a few ulp through atan2 / asin / division by the pixel size), so a tighter check fails on noise, not on the interpolant
#define PXL_TILED_TOL 1e-10
#ifndef PXL_TNX
#define PXL_TNX 7          // lattice nodes along a tile's width (degree 6)
#endif
#ifndef PXL_TNY
#define PXL_TNY 6
#endif
#define PXL_TCHK0 (PXL_TNX * PXL_TNY <= 40 ? 40 : 48)     // first of the check lanes of a tile's wave
#define PXL_TNCHK 12                                       // check points per tile (lanes PXL_TCHK0 ... + 11)
#ifndef PXL_TH
#define PXL_TH 32          // tile height (output rows)
#endif
#ifndef PXL_TW
#define PXL_TW 128         // tile width (output columns): 64 or 128 (one or two waves side by side in the 256-thread block)
#endif
#define PXL_TROWS (256 / PXL_TW)        // output rows a block covers at once
// Lagrange basis on the N equispaced nodes 0, h, ..., (N-1) h.  The denominators prod_{c != a} (a - c) h are
// constants (+-120, 24, 12 x h^5 for N = 6; 24, 6, 4 x h^4 for N = 5): no division in the kernel.
template <int N>
__device__ inline void lagrange_weights(double u, double h, double* w) {
    double d[N];
#pragma unroll
    for (int c = 0; c < N; ++c) d[c] = u - c * h;
    double hp = 1.0;
#pragma unroll
    for (int c = 1; c < N; ++c) hp *= h;
    const double rh = 1.0 / hp;                                   // one division per call (h^(N-1))
#pragma unroll
    for (int a = 0; a < N; ++a) {
        double num = 1.0;
        int k = 1;                                                // prod_{c != a} (a - c), an integer
#pragma unroll
        for (int c = 0; c < N; ++c)
            if (c != a) { num *= d[c]; k *= (a - c); }
        w[a] = num * (rh / (double)k);
    }
}
// Three launches.  k_generic_lattice: every tile's 42 lattice points + 12 check points, all tiles in parallel (one wave
// per tile; the ~10 libm calls per point are the expensive part and no pixel waits behind them); per tile it leaves the
// lattice coordinates and a flag (1 = the interpolant failed its check or a point is non-finite / not visible).
// k_reproject_generic_tiled: one block per tile, 16 pixels per thread, no LDS and no barrier: the tile's lattice is
// wave-uniform data.  k_reproject_generic_exact_tiles: the flagged tiles, per pixel.
// The pixels of a tile sit at integer positions 0..127 x 0..31 of the lattice's coordinate system, the same in every
// tile: their Lagrange weights are compile-time tables.
struct TileWeights { double wx[PXL_TW][PXL_TNX]; double wy[PXL_TH][PXL_TNY]; };
constexpr double lag_w(int n, double h, double u, int a) {
    double num = 1.0, den = 1.0;
    for (int c = 0; c < n; ++c)
        if (c != a) { num *= (u - c * h); den *= ((a - c) * h); }
    return num / den;
}
constexpr TileWeights make_tile_weights() {
    TileWeights t{};
    for (int u = 0; u < PXL_TW; ++u)
        for (int a = 0; a < PXL_TNX; ++a) t.wx[u][a] = lag_w(PXL_TNX, (PXL_TW - 1.0) / (PXL_TNX - 1), (double)u, a);
    for (int v = 0; v < PXL_TH; ++v)
        for (int b = 0; b < PXL_TNY; ++b) t.wy[v][b] = lag_w(PXL_TNY, (PXL_TH - 1.0) / (PXL_TNY - 1), (double)v, b);
    return t;
}
__constant__ TileWeights c_tile_weights = make_tile_weights();

__global__ __launch_bounds__(256) void k_generic_lattice(GenericParams p, int64_t ntx, int64_t ntiles,
                                                         double2* __restrict__ lat, int32_t* __restrict__ flag) {
    __shared__ double lx[4][PXL_TNX * PXL_TNY], ly[4][PXL_TNX * PXL_TNY];
    __shared__ int bad[4];
    const int w = threadIdx.x >> 6, k = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * 4 + w;
    const bool live = tile < ntiles;
    const int64_t ti0 = live ? (tile % ntx) * PXL_TW : 0, tj0 = live ? (tile / ntx) * PXL_TH : 0;
    const double hx = (PXL_TW - 1.0) / (PXL_TNX - 1), hy = (PXL_TH - 1.0) / (PXL_TNY - 1);
    if (k == 0) bad[w] = 0;
    __syncthreads();
    double ex = 0.0, ey = 0.0, cu = 0.0, cv = 0.0;
    bool is_check = false;
    if (live && k < PXL_TNX * PXL_TNY) {
        const int a = k % PXL_TNX, b = k / PXL_TNX;
        bool vis;
        generic_coords(p, (double)(ti0 + 1) + a * hx, (double)(tj0 + 1) + b * hy, &ex, &ey, &vis);
        lx[w][k] = ex; ly[w][k] = ey;
        lat[tile * (PXL_TNX * PXL_TNY) + k] = make_double2(ex, ey);
        if (!vis || !isfinite(ex) || !isfinite(ey)) bad[w] = 1;
    } else if (live && k >= PXL_TCHK0 && k < PXL_TCHK0 + PXL_TNCHK) {
        // off-lattice check points, spread over the tile (fractions of its extent; none coincides with a node)
        const double uf[12] = {0.10, 0.50, 0.90, 0.30, 0.70, 0.95, 0.04, 0.21, 0.41, 0.59, 0.79, 0.985};
        const double vf[12] = {0.12, 0.47, 0.82, 0.70, 0.23, 0.94, 0.55, 0.97, 0.05, 0.88, 0.35, 0.63};
        const double us[12] = {uf[0] * (PXL_TW - 1), uf[1] * (PXL_TW - 1), uf[2] * (PXL_TW - 1), uf[3] * (PXL_TW - 1), uf[4] * (PXL_TW - 1),
                               uf[5] * (PXL_TW - 1), uf[6] * (PXL_TW - 1), uf[7] * (PXL_TW - 1), uf[8] * (PXL_TW - 1), uf[9] * (PXL_TW - 1),
                               uf[10] * (PXL_TW - 1), uf[11] * (PXL_TW - 1)};
        const double vs[12] = {vf[0] * (PXL_TH - 1), vf[1] * (PXL_TH - 1), vf[2] * (PXL_TH - 1), vf[3] * (PXL_TH - 1), vf[4] * (PXL_TH - 1),
                               vf[5] * (PXL_TH - 1), vf[6] * (PXL_TH - 1), vf[7] * (PXL_TH - 1), vf[8] * (PXL_TH - 1), vf[9] * (PXL_TH - 1),
                               vf[10] * (PXL_TH - 1), vf[11] * (PXL_TH - 1)};
        cu = us[k - PXL_TCHK0]; cv = vs[k - PXL_TCHK0];
        bool vis;
        generic_coords(p, (double)(ti0 + 1) + cu, (double)(tj0 + 1) + cv, &ex, &ey, &vis);
        is_check = true;
        if (!vis || !isfinite(ex) || !isfinite(ey)) bad[w] = 1;
    }
    __syncthreads();
    if (is_check && !bad[w]) {
        double wx[PXL_TNX], wy[PXL_TNY];
        lagrange_weights<PXL_TNX>(cu, hx, wx);
        lagrange_weights<PXL_TNY>(cv, hy, wy);
        double sx = 0.0, sy = 0.0;
        for (int b = 0; b < PXL_TNY; ++b)
            for (int a = 0; a < PXL_TNX; ++a) {
                const double ww = wx[a] * wy[b];
                sx = __builtin_fma(ww, lx[w][b * PXL_TNX + a], sx);
                sy = __builtin_fma(ww, ly[w][b * PXL_TNX + a], sy);
            }
        if (!(fabs(sx - ex) <= PXL_TILED_TOL && fabs(sy - ey) <= PXL_TILED_TOL)) bad[w] = 1;
    }
    __syncthreads();
    if (live && k == 0) {
        flag[tile] = bad[w];
        if (bad[w] && p.exact_tiles) atomicAdd(p.exact_tiles, 1u);
    }
}
// the tiles whose interpolant failed its check: exact evaluation per pixel (a launch of its own, so that the hot
// kernel below carries no libm code; blocks of tiles that passed exit at once)
// A small fixed grid that walks the tiles: when no tile failed (the counter of the lattice launch is zero) every block leaves at
// once -- 2 us instead of the 4.2 us that one (empty) block per tile cost on a 4096^2 patch.
__global__ __launch_bounds__(256) void k_reproject_generic_exact_tiles(GenericParams p, const int32_t* __restrict__ flag, int64_t gx, int64_t ntiles) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && p.exact_tiles_next) *p.exact_tiles_next = 0u;
    if (p.exact_tiles && *p.exact_tiles == 0u) return;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        if (!flag[tile]) continue;
        const int64_t bx = tile % gx, by = tile / gx;
        const int64_t i = bx * PXL_TW + (threadIdx.x & (PXL_TW - 1)), jr0 = by * PXL_TH + threadIdx.x / PXL_TW;
        if (i >= p.nxo) continue;
        for (int q = 0; q < PXL_TH / PXL_TROWS; ++q) {
            const int64_t jr = jr0 + PXL_TROWS * q;
            if (jr < p.nyo) {
                double x, y; bool visible;
                generic_coords(p, (double)(i + 1), (double)(jr + 1), &x, &y, &visible);
                generic_store(p, jr * p.nxo + i, x, y, visible);
            }
        }
    }
}
__global__ __launch_bounds__(256) void k_reproject_generic_tiled(GenericParams p, const double2* __restrict__ lat,
                                                                 const int32_t* __restrict__ flag) {
    const int64_t tile = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
    const int64_t ti0 = (int64_t)blockIdx.x * PXL_TW, tj0 = (int64_t)blockIdx.y * PXL_TH;   // 0-based tile origin
    const int tid = threadIdx.x;
    // ry through readfirstlane: the row weights c_tile_weights.wy[ry + 4q] are then wave-uniform SCALAR loads; as plain
    // tid >> 6 they were three 16-byte VECTOR loads per pixel, a dependent L1 round trip in front of every pixel's taps
    const int cx = tid & (PXL_TW - 1), ry = __builtin_amdgcn_readfirstlane(tid / PXL_TW);
    const int64_t i = ti0 + cx;
    if (i >= p.nxo) return;
    if (flag[tile]) return;                      // k_reproject_generic_exact_tiles does this tile
    const double2* L = lat + tile * (PXL_TNX * PXL_TNY);
    double wx[PXL_TNX];
#pragma unroll
    for (int a = 0; a < PXL_TNX; ++a) wx[a] = c_tile_weights.wx[cx][a];
    // column-interpolated lattice: one value per lattice row, reused by this thread's four rows
    double colx[PXL_TNY], coly[PXL_TNY];
#pragma unroll
    for (int b = 0; b < PXL_TNY; ++b) {
        double sx = 0.0, sy = 0.0;
#pragma unroll
        for (int a = 0; a < PXL_TNX; ++a) { const double2 v = L[b * PXL_TNX + a]; sx = __builtin_fma(wx[a], v.x, sx); sy = __builtin_fma(wx[a], v.y, sy); }
        colx[b] = sx; coly[b] = sy;
    }
#pragma unroll
    for (int q = 0; q < PXL_TH / PXL_TROWS; ++q) {
        const int r = ry + PXL_TROWS * q;
        const int64_t jr = tj0 + r;
        if (jr < p.nyo) {
            double x = 0.0, y = 0.0;
#pragma unroll
            for (int b = 0; b < PXL_TNY; ++b) { const double wy = c_tile_weights.wy[r][b]; x = __builtin_fma(wy, colx[b], x); y = __builtin_fma(wy, coly[b], y); }
            generic_store(p, jr * p.nxo + i, x, y, true);
        }
    }
}

// Round 4, third form: one pixel per lane as in round 3's kernel (24 registers of column data, 6 waves per SIMD) with a LEAN
// interior path.  Round 3's loop costs ~99 VALU instructions per pixel (26 M wave instructions per 4096^2 patch,
// profiles/r02_tan_mosaic_counters.txt) of which only 12 + 11 are the coordinate interpolation and the blend: the rest is 64-bit
// index arithmetic, range tests, two isfinite classifications, clamps and per-tap selects -- on a kernel whose FP64-rate
// instructions alone take 48 us of its 76 us.  Here:
//   * the cell is floor -> subtract -> v_cvt_i32_f64, which SATURATES out-of-range values and turns NaN into 0: both then fail the
//     one unsigned range test per axis, so neither clamp nor isfinite is needed on the fast path;
//   * "all four taps inside, no seam" is (unsigned)(i0 - 1) < nx - 1 and (unsigned)(j0 - 1) < ny - 1;
//   * the offset is 32-bit when the plane has fewer than 2^31 elements (wave-uniform choice), one 64-bit shift-add per row address;
//   * the two taps of a row are one 16-byte load (8-byte aligned);
//   * rows that fail the test anywhere in the wave (seam, edges, horizon, NaN) are redone afterwards by generic_store, one
//     non-unrolled copy of the general code.
// Same operations on the same operands as generic_store's interior path: the bits are those of round 3's kernel.
// Measured and dropped on top of this form (round 4, 16-patch mosaic, one box): 16-byte stores from one-pixel lanes by a DPP swap
// between lanes 2k and 2k+1 (half the store instructions: 1.438 against 1.357 ms -- slower); the exact fallback folded into the
// lattice launch so that the third launch disappears (1.357 against 1.355 ms: nothing, and 112 VGPRs in the lattice kernel).
__global__ __launch_bounds__(256) void k_reproject_generic_tiled3(GenericParams p, const double2* __restrict__ lat,
                                                                  const int32_t* __restrict__ flag) {
    const int64_t tile = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
    const int64_t ti0 = (int64_t)blockIdx.x * PXL_TW, tj0 = (int64_t)blockIdx.y * PXL_TH;   // 0-based tile origin
    const int tid = threadIdx.x;
    const int cx = tid & (PXL_TW - 1), ry = __builtin_amdgcn_readfirstlane(tid / PXL_TW);
    const int64_t i = ti0 + cx;
    if (i >= p.nxo) return;
    if (flag[tile]) return;                      // k_reproject_generic_exact_tiles does this tile
    const double2* L = lat + tile * (PXL_TNX * PXL_TNY);
    double wx[PXL_TNX];
#pragma unroll
    for (int a = 0; a < PXL_TNX; ++a) wx[a] = c_tile_weights.wx[cx][a];
    double colx[PXL_TNY], coly[PXL_TNY];
#pragma unroll
    for (int b = 0; b < PXL_TNY; ++b) {
        double sx = 0.0, sy = 0.0;
#pragma unroll
        for (int a = 0; a < PXL_TNX; ++a) { const double2 v = L[b * PXL_TNX + a]; sx = __builtin_fma(wx[a], v.x, sx); sy = __builtin_fma(wx[a], v.y, sy); }
        colx[b] = sx; coly[b] = sy;
    }
    auto coords = [&](int r, double* x, double* y) {
        double sx = 0.0, sy = 0.0;
#pragma unroll
        for (int b = 0; b < PXL_TNY; ++b) { const double wy = c_tile_weights.wy[r][b]; sx = __builtin_fma(wy, colx[b], sx); sy = __builtin_fma(wy, coly[b], sy); }
        *x = sx; *y = sy;
    };
    const int64_t total = p.nxo * p.nyo;
    const uint32_t nxm1 = (uint32_t)(p.nx - 1), nym1 = (uint32_t)(p.ny - 1);
    const int32_t nx32 = (int32_t)p.nx;
    const bool small = p.nx * p.ny < 0x7fffffffLL && p.nx < 0x7fffffffLL;            // wave-uniform: 32-bit element offsets
    struct __attribute__((packed, aligned(8))) Pair { double a, b; };
    uint32_t slow_rows = 0;
    constexpr int NQ = PXL_TH / PXL_TROWS;
    // rows in groups of G: the group's 2 G tap loads are issued back to back before the first blend (a row at a time the wave had
    // two loads in flight and sat in s_waitcnt for 75 % of its cycles, round 3's counters)
#ifndef PXL_T3_G
#define PXL_T3_G 4
#endif

    constexpr int G = PXL_T3_G;
#pragma unroll
    for (int q0 = 0; q0 < NQ; q0 += G) {
        double fx[G], fy[G];
        int64_t off[G];
        bool ok = true;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int r = ry + PXL_TROWS * (q0 + g);
            double x, y;
            coords(r, &x, &y);
            const double flx = floor(x), fly = floor(y);
            fx[g] = x - flx; fy[g] = y - fly;
            const int32_t i0 = (int32_t)flx, j0 = (int32_t)fly;                        // saturating; NaN -> 0
            ok = ok && (uint32_t)(i0 - 1) < nxm1 && (uint32_t)(j0 - 1) < nym1;
            off[g] = small ? (int64_t)((j0 - 1) * nx32 + (i0 - 1)) : (int64_t)(j0 - 1) * p.nx + (i0 - 1);
        }
        const bool inside = tj0 + ry + PXL_TROWS * (q0 + G - 1) < p.nyo;               // wave-uniform: the whole group is inside the map
        if (inside && __all(ok)) {
            for (int c = 0; c < p.nc; ++c) {
                const double* pl = p.src + (int64_t)c * p.nx * p.ny;
                Pair tp[G], bt[G];
#pragma unroll
                for (int g = 0; g < G; ++g) { tp[g] = *reinterpret_cast<const Pair*>(pl + off[g]); bt[g] = *reinterpret_cast<const Pair*>(pl + off[g] + p.nx); }
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const double top = (1 - fx[g]) * tp[g].a + fx[g] * tp[g].b;
                    const double bot = (1 - fx[g]) * bt[g].a + fx[g] * bt[g].b;
                    // non-temporal: the output is never read back here, and kept out of the caches it leaves them to the taps that
                    // neighbouring rows and tiles share (mosaic of 16 patches, plans reused: 1.10 -> 1.04-1.05 ms; rows in groups of 2 or
                    // 8, a rolled row loop: no better; tiles handed out so that each XCD covers a contiguous eighth of the patch:
                    // 1.075 against 1.056 ms; right taps taken from the neighbouring lane by DPP where it starts one cell further, own
                    // 8-byte loads elsewhere -- half the bytes through the texture addresser: 1.24-1.57 against 1.06 ms, the masked
                    // second round of loads costs more than the pair loads' overlap)
                    __builtin_nontemporal_store((1 - fy[g]) * top + fy[g] * bot, &p.dst[(int64_t)c * total + (tj0 + ry + PXL_TROWS * (q0 + g)) * p.nxo + i]);
                }
            }
        } else {
#pragma unroll
            for (int g = 0; g < G; ++g)
                if (tj0 + ry + PXL_TROWS * (q0 + g) < p.nyo) slow_rows |= 1u << (q0 + g);
        }
    }
#pragma unroll 1
    while (slow_rows) {
        const int q = __builtin_ctz(slow_rows);
        slow_rows &= slow_rows - 1;
        const int r = ry + PXL_TROWS * q;
        double x, y;
        // (compile-time row index for the weights table is lost here: a plain loop over the table)
        double sx = 0.0, sy = 0.0;
#pragma unroll
        for (int b = 0; b < PXL_TNY; ++b) { const double wy = c_tile_weights.wy[r][b]; sx = __builtin_fma(wy, colx[b], sx); sy = __builtin_fma(wy, coly[b], sy); }
        x = sx; y = sy;
        generic_store(p, (tj0 + r) * p.nxo + i, x, y, true);
    }
}

// ---- scattered sample: fused sky2pix!(safe=true) [car_proj.jl:165-193] + 2x2 gather + lerp.
// An irregular gather: each point touches two 16-byte spans in two different rows of a multi-GB map, so
// the kernel is bound by random-sector fetches, not by bytes.  Each lane handles PXL_SUNR points per trip
// and issues all their taps before any arithmetic (4x the gathers in flight per lane); tap indices are
// 32-bit and the RA wrap is one conditional add/subtract (safe sky2pix keeps x within half a period of the
// map centre), with the oracle's full modulo kept only as the out-of-range path.
#ifndef PXL_SUNR
#define PXL_SUNR 4
#endif
// sky2pix(safe=true) of one point for the samplers.  The branch-free rewind covers every finite input with a sane
// period (bit-identical to rewind(), see pxl_device.h); the library-fmod form is kept OUT of line, so that the
// row-pair kernel carries one copy of it instead of eight inlined ones (either form runs it at the same speed; the direct
// kernel keeps the inlined form, see there).
// (arguments and result in registers: with the struct passed by value every lane stored its 144 bytes to scratch on
// every trip, fast path or not -- 8.5 GB of HBM writes per 1e9 points in WRITE_SIZE)
__device__ __noinline__ double2 sample_coords_slow(double a, double d, double ia0, double a0, double rda, double px, double cx,
                                                   double rpx, double id0, double d0, double rdd, double py, double cy, double rpy) {
    // safe = 1, reciprocal form (the only one the samplers use): s2p_x / s2p_y of pxl_device.h with the library fmod
    return make_double2(rewind(ia0 + (a - a0) * rda, px, cx, rpx), rewind(id0 + (d - d0) * rdd, py, cy, rpy));
}
__device__ inline void sample_coords(const Sky2Pix& s, double a, double d, double* x, double* y) {
    if (s.safe && s.form != PXL_FORM_DIV) {
        bool ok0, ok1;
        *x = rewind_try(s.c.ia0 + (a - s.c.a0) * s.rda, s.px, s.cx, s.rpx, &ok0);
        *y = rewind_try(s.c.id0 + (d - s.c.d0) * s.rdd, s.py, s.cy, s.rpy, &ok1);
        if (__builtin_expect(!(ok0 && ok1), 0)) {
            const double2 r = sample_coords_slow(a, d, s.c.ia0, s.c.a0, s.rda, s.px, s.cx, s.rpx, s.c.id0, s.c.d0, s.rdd, s.py, s.cy, s.rpy);
            *x = r.x; *y = r.y;
        }
        return;
    }
    *x = s2p_x(s, a); *y = s2p_y(s, d);
}
template <typename T>
__global__ __launch_bounds__(256) void k_sample_bilinear(Sky2Pix s, const T* __restrict__ src, int64_t nx,
                                                         int64_t ny, int32_t nc, int64_t row0, int64_t nrows,
                                                         int periodic, int64_t n, const double2* __restrict__ sky,
                                                         T* __restrict__ out) {
    const int64_t chunk = (int64_t)blockDim.x * PXL_SUNR;
    const int64_t plane = nx * nrows;
    for (int64_t k0 = (int64_t)blockIdx.x * chunk + threadIdx.x; k0 < n; k0 += (int64_t)gridDim.x * chunk) {
        double2 ad[PXL_SUNR];
#pragma unroll
        for (int u = 0; u < PXL_SUNR; ++u) {
            int64_t k = k0 + u * blockDim.x;
            ad[u] = (k < n) ? sky[k] : make_double2(0.0, 0.0);
        }
        int64_t o00[PXL_SUNR], o10[PXL_SUNR], o01[PXL_SUNR], o11[PXL_SUNR];   // element offsets, -1 = reads as 0
        double fx[PXL_SUNR], fy[PXL_SUNR];
        bool fin[PXL_SUNR];
#pragma unroll
        for (int u = 0; u < PXL_SUNR; ++u) {
            // inlined evaluators here (fast path + library fmod per coordinate): the out-of-line fallback of
            // sample_coords() buys this kernel a fourth wave per SIMD and costs it 15 % (48.6-50.2 vs 55.9-57.3 ms per 1e9
            // points, same box) -- with four separate taps per point more waves in flight evict each other's sectors
            double x = s2p_x(s, ad[u].x), y = s2p_y(s, ad[u].y);
            fin[u] = isfinite(x) && isfinite(y);
            int32_t i0, j0;
            split_cell(x, &i0, &fx[u]);
            split_cell(y, &j0, &fy[u]);
            int64_t ia = i0, ib = (int64_t)i0 + 1;
            bool oka = true, okb = true;
            if (periodic) { ia = wrap_col(ia, nx); ib = wrap_col(ib, nx); }
            else { oka = (ia >= 1 && ia <= nx); okb = (ib >= 1 && ib <= nx); }
            int64_t ja = (int64_t)j0 - 1 - row0, jb = ja + 1;                    // resident row indices
            bool rowa = (j0 >= 1 && j0 <= ny && ja >= 0 && ja < nrows);
            bool rowb = ((int64_t)j0 + 1 >= 1 && (int64_t)j0 + 1 <= ny && jb >= 0 && jb < nrows);
            o00[u] = (rowa && oka) ? ja * nx + (ia - 1) : -1;
            o10[u] = (rowa && okb) ? ja * nx + (ib - 1) : -1;
            o01[u] = (rowb && oka) ? jb * nx + (ia - 1) : -1;
            o11[u] = (rowb && okb) ? jb * nx + (ib - 1) : -1;
        }
        for (int c = 0; c < nc; ++c) {
            const T* pl = src + (int64_t)c * plane;
            double m00[PXL_SUNR], m10[PXL_SUNR], m01[PXL_SUNR], m11[PXL_SUNR];
            // interior points (all four taps on the map, the two columns adjacent): ONE 2-element load per row (element-
            // aligned only; 54.9 vs 57.8 ms per 1e9 points against four separate taps, same box); the rest (seam, edges, rows
            // outside the window) take the four taps in a rare branch
            struct __attribute__((packed, aligned(sizeof(T)))) TT { T a, b; };
            bool wide[PXL_SUNR];
#pragma unroll
            for (int u = 0; u < PXL_SUNR; ++u) {
                wide[u] = o00[u] >= 0 && o01[u] >= 0 && o10[u] == o00[u] + 1 && o11[u] == o01[u] + 1;
                // the other points still issue the load (no branch in front of the gathers), from an address that always
                // exists: the first coordinate pair of the batch (16 readable bytes whenever n >= 1).  The map itself may
                // not have two elements to read -- an empty resident window (src == NULL) or a 1 x 1 one
                const TT ra = *(wide[u] ? reinterpret_cast<const TT*>(pl + o00[u]) : reinterpret_cast<const TT*>(sky));
                const TT rb = *(wide[u] ? reinterpret_cast<const TT*>(pl + o01[u]) : reinterpret_cast<const TT*>(sky));
                m00[u] = (double)ra.a; m10[u] = (double)ra.b; m01[u] = (double)rb.a; m11[u] = (double)rb.b;
            }
#pragma unroll
            for (int u = 0; u < PXL_SUNR; ++u) {
                if (__builtin_expect(!wide[u], 0)) {
                    m00[u] = o00[u] >= 0 ? (double)pl[o00[u]] : 0.0;
                    m10[u] = o10[u] >= 0 ? (double)pl[o10[u]] : 0.0;
                    m01[u] = o01[u] >= 0 ? (double)pl[o01[u]] : 0.0;
                    m11[u] = o11[u] >= 0 ? (double)pl[o11[u]] : 0.0;
                }
            }
#pragma unroll
            for (int u = 0; u < PXL_SUNR; ++u) {
                int64_t k = k0 + u * blockDim.x;
                double top = (1 - fx[u]) * m00[u] + fx[u] * m10[u];
                double bot = (1 - fx[u]) * m01[u] + fx[u] * m11[u];
                double v = (1 - fy[u]) * top + fy[u] * bot;
                if (k < n) out[(int64_t)c * n + k] = (T)(fin[u] ? v : __builtin_nan(""));
            }
        }
    }
}

// ---- row-pair layout for scattered sampling.  The gather above is bound by random 64-byte sector fetches (about
// 2.25 per point: the two taps of a row are neighbours, the two rows are nx elements apart).  In the pair layout
// element (p, i) holds (v[p-1][i], v[p][i]) -- rows outside the resident window read as zero, p = 0 .. nrows -- so
// the whole 2x2 neighbourhood of a point is two ADJACENT 2-element entries, kept inside one sector by the grouping
// below: one random sector per point, for 8/3 of the map's footprint (288 GB of HBM is there to be used) and one
// streaming pass to build it.  Same taps, same arithmetic: results are bit-identical to k_sample_bilinear.
// What bounds it (tools/research/exp_random_reach.cpp, exp_scalar_gather.cpp): the memory system serves ~54 G random
// 64-byte requests per second whatever the footprint (128 MiB ... 48 GiB, Infinity Cache or HBM, vector and scalar
// path together), and ~38 G cells/s when the coordinate and result streams share it; this kernel reaches 31-33.
template <typename T> struct Vec2T;
template <> struct Vec2T<double> { typedef double2 type; };
template <> struct Vec2T<float>  { typedef float2 type; };

// Sector grouping: the entries of a pair row are stored in groups of one 64-byte sector (E = 4 Float64 or 8 Float32
// entries) that OVERLAP by one entry: group s holds columns s*(E-1)+1 ... s*(E-1)+E (1-based; columns past nx wrap
// round to 1, 2, ... -- the sampler zeroes them on a map that is not periodic).  A cell's two adjacent entries therefore
// always share a sector: 1.0 random sector per point instead of 1.25 (Float64) / 1.125 (Float32), for 4/3 (8/7) of the
// plain row-pair footprint.  Measured with the cells forced into one sector: 33.3 -> 30.2 ms per 1e9 points.
template <typename T> struct PairGroup {
    static constexpr int E = 64 / (2 * (int)sizeof(T));     // entries per sector
    static constexpr int C = E - 1;                         // cells (distinct left columns) per sector
    __host__ __device__ static inline int64_t groups(int64_t nx) { return (nx + C - 1) / C; }
};
template <typename T>
__global__ __launch_bounds__(256) void k_build_rowpairs(const T* __restrict__ src, int64_t nx, int64_t nrows,
                                                        typename Vec2T<T>::type* __restrict__ pairs, int fronts) {
    // block = (256 entries of a pair row, PXL_POS_ROWS pair rows, component): a lane walks down its column carrying the
    // row above; stores are contiguous 2-element entries, loads run along the row with every (E-1)-th column read twice
    typedef typename Vec2T<T>::type T2;
    constexpr int E = PairGroup<T>::E, C = PairGroup<T>::C;
    const int64_t pitch = PairGroup<T>::groups(nx) * E;               // entries per pair row
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= pitch) return;
    const int64_t g = t / E, e = t - g * E;
    int64_t i = g * C + e;                                              // 0-based column
    if (i >= nx) i %= nx;
    const int64_t c = blockIdx.z;
    const T* pl = src + c * nx * nrows;
    T2* out = pairs + c * pitch * (nrows + 1);
    // the copy writes 8/3 of what it reads: like the other write-heavy kernels it deals its row tiles into `fronts` parts of the
    // destination (tile row y works in part y % fronts) -- several write fronts instead of one
    const int64_t per = (gridDim.y + fronts - 1) / fronts;
    const int64_t yy = fronts > 1 ? (int64_t)(blockIdx.y % fronts) * per + blockIdx.y / fronts : (int64_t)blockIdx.y;
    const int64_t p0 = yy * PXL_POS_ROWS;
    if (p0 > nrows) return;
    const int64_t p1 = (p0 + PXL_POS_ROWS <= nrows) ? p0 + PXL_POS_ROWS : nrows + 1;     // pair rows [p0, p1)
    T above = (p0 >= 1) ? pl[(p0 - 1) * nx + i] : (T)0;
    for (int64_t p = p0; p < p1; ++p) {
        const T here = (p < nrows) ? pl[p * nx + i] : (T)0;
        T2 v; v.x = above; v.y = here;
        out[p * pitch + t] = v;
        above = here;
    }
}

// points per lane per trip of k_sample_pairs: the kernel is bound by the random-request rate of the memory system, not
// by loads in flight -- Float64: 2, 4 and 8 within 3 %; Float32 entries: 29.8 / 28.9 / 27.8 ms per 1e9 points with 1 / 2 / 4
template <typename T> struct PairsUnroll { static constexpr int value = PXL_SUNR; };
template <typename T>
__global__ __launch_bounds__(256) void k_sample_pairs(Sky2Pix s, const typename Vec2T<T>::type* __restrict__ pairs,
                                                      int64_t nx, int64_t ny, int32_t nc, int64_t row0, int64_t nrows,
                                                      int periodic, int64_t n, const double2* __restrict__ sky,
                                                      T* __restrict__ out) {
    typedef typename Vec2T<T>::type T2;
    constexpr int SUNR = PairsUnroll<T>::value;
    constexpr int E = PairGroup<T>::E, C = PairGroup<T>::C;
    const int64_t chunk = (int64_t)blockDim.x * SUNR;
    const int64_t pitch = PairGroup<T>::groups(nx) * E;
    const int64_t plane = pitch * (nrows + 1);
    for (int64_t k0 = (int64_t)blockIdx.x * chunk + threadIdx.x; k0 < n; k0 += (int64_t)gridDim.x * chunk) {
        double2 ad[SUNR];
#pragma unroll
        for (int u = 0; u < SUNR; ++u) {
            int64_t k = k0 + u * blockDim.x;
            ad[u] = (k < n) ? sky[k] : make_double2(0.0, 0.0);
        }
        int64_t oa[SUNR], ob[SUNR];                  // pair-entry offsets of columns i0 and i0 + 1, -1 = zeros
        double fx[SUNR], fy[SUNR];
        bool fin[SUNR];
#pragma unroll
        for (int u = 0; u < SUNR; ++u) {
            double x, y;
            sample_coords(s, ad[u].x, ad[u].y, &x, &y);
            fin[u] = isfinite(x) && isfinite(y);
            int32_t i0, j0;
            split_cell(x, &i0, &fx[u]);
            split_cell(y, &j0, &fy[u]);
            int64_t ia = i0, ib = (int64_t)i0 + 1;
            bool oka = true, okb = true;
            if (periodic) { ia = wrap_col(ia, nx); ib = wrap_col(ib, nx); }
            else { oka = (ia >= 1 && ia <= nx); okb = (ib >= 1 && ib <= nx); }
            const int64_t p = (int64_t)j0 - row0;            // entry p holds resident rows p-1 (cell row j0) and p
            const bool rows = (p >= 0 && p <= nrows);
            // entry of column c (1-based, <= nx) in its own group: (c-1)/C * E + (c-1)%C; the right-hand column of a cell
            // is the next entry of the SAME group (the wrapped column 1 after column nx included)
            const uint32_t ca = (uint32_t)((oka ? ia : ib) - 1), ga = ca / (uint32_t)C;
            const int64_t ent = p * pitch + (int64_t)ga * E + (ca - ga * (uint32_t)C);
            oa[u] = (rows && oka) ? ent : -1;
            ob[u] = (rows && okb) ? (oka ? ent + 1 : ent) : -1;
        }
        for (int c = 0; c < nc; ++c) {
            const T2* pl = pairs + (int64_t)c * plane;
            T2 a[SUNR], b[SUNR];
#pragma unroll
            for (int u = 0; u < SUNR; ++u) {
                // unconditional whole-entry loads (a conditional one became four 8-byte FLAT loads through a select
                // between the entry and a zero on the stack); an off-map entry reads entry 0 and is zeroed afterwards
                a[u] = pl[oa[u] >= 0 ? oa[u] : 0];
                b[u] = pl[ob[u] >= 0 ? ob[u] : 0];
            }
#pragma unroll
            for (int u = 0; u < SUNR; ++u) {
                if (oa[u] < 0) { a[u].x = (T)0; a[u].y = (T)0; }
                if (ob[u] < 0) { b[u].x = (T)0; b[u].y = (T)0; }
            }
#pragma unroll
            for (int u = 0; u < SUNR; ++u) {
                int64_t k = k0 + u * blockDim.x;
                double top = (1 - fx[u]) * (double)a[u].x + fx[u] * (double)b[u].x;
                double bot = (1 - fx[u]) * (double)a[u].y + fx[u] * (double)b[u].y;
                double v = (1 - fy[u]) * top + fy[u] * bot;
                if (k < n) out[(int64_t)c * n + k] = (T)(fin[u] ? v : __builtin_nan(""));
            }
        }
    }
}
