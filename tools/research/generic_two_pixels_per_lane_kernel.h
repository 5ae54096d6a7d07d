// generic_two_pixels_per_lane_kernel.h -- NOT COMPILED INTO THE LIBRARY.  The first round-4 form of the CAR<->Gnomonic pixel kernel: two
// adjacent output pixels per lane, 16-byte taps and 16-byte stores.  105 VGPRs (4 waves per SIMD against 7) and 4 loads in flight per
// wave: 1.711 ms on the 16-patch mosaic against round 3's 1.62-1.64 ms and k_reproject_generic_tiled3's 1.33-1.36 ms.  Bit-identical.

// Round 4: the same tile, TWO ADJACENT OUTPUT PIXELS PER LANE and 16-byte memory instructions.  The round-3 kernel above gives a
// lane one pixel: its four taps are four 8-byte loads and its result an 8-byte store, and the texture addresser -- which walks a
// wave's 64 addresses at the same pace whether they carry 8 or 16 bytes (22 L1 accesses per tap instruction,
// profiles/r02_tan_mosaic_counters.txt) -- was busy 86 of the launch's 108 us.  Here a pixel's two taps of one source row are ONE
// 16-byte load (cells i0, i0 + 1 are adjacent in memory; 8-byte aligned is all a global dwordx4 load needs), and a lane stores its
// two pixels with one 16-byte store: a quarter of the tap instructions and half of the store instructions per pixel.  The seam of
// a periodic source (i0 = 0 or nx), map edges, invisible and non-finite points take the per-pixel path of generic_store: same
// operations on the same operands, so the bits do not depend on the path.  A wave covers one 128-pixel row of the tile at a time
// (wave w of the block: rows w, w + 4, ...), its lattice data is wave-uniform.
struct __attribute__((packed, aligned(8))) PxlPair { double a, b; };
__global__ __launch_bounds__(256) void k_reproject_generic_tiled2(GenericParams p, const double2* __restrict__ lat,
                                                                  const int32_t* __restrict__ flag) {
    const int64_t tile = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
    const int64_t ti0 = (int64_t)blockIdx.x * PXL_TW, tj0 = (int64_t)blockIdx.y * PXL_TH;   // 0-based tile origin
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cx = 2 * lane;
    const int64_t i = ti0 + cx;
    if (i >= p.nxo) return;
    if (flag[tile]) return;                      // k_reproject_generic_exact_tiles does this tile
    const bool two = i + 1 < p.nxo;
    const double2* L = lat + tile * (PXL_TNX * PXL_TNY);
    // column-interpolated lattice of the lane's two columns: one value per lattice row, reused by all of the lane's rows
    double colx[2][PXL_TNY], coly[2][PXL_TNY];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        double wx[PXL_TNX];
#pragma unroll
        for (int a = 0; a < PXL_TNX; ++a) wx[a] = c_tile_weights.wx[cx + e][a];
#pragma unroll
        for (int b = 0; b < PXL_TNY; ++b) {
            double sx = 0.0, sy = 0.0;
#pragma unroll
            for (int a = 0; a < PXL_TNX; ++a) { const double2 v = L[b * PXL_TNX + a]; sx = __builtin_fma(wx[a], v.x, sx); sy = __builtin_fma(wx[a], v.y, sy); }
            colx[e][b] = sx; coly[e][b] = sy;
        }
    }
    const int64_t total = p.nxo * p.nyo;
    const bool vec_store = two && ((p.nxo & 1) == 0) && (((uintptr_t)p.dst & 15) == 0);
    auto coords = [&](int r, int e, double* x, double* y) {
        double sx = 0.0, sy = 0.0;
#pragma unroll
        for (int b = 0; b < PXL_TNY; ++b) { const double wy = c_tile_weights.wy[r][b]; sx = __builtin_fma(wy, colx[e][b], sx); sy = __builtin_fma(wy, coly[e][b], sy); }
        *x = sx; *y = sy;
    };
    uint32_t slow_rows = 0;                      // wave-uniform: rows of this wave that could not take the fast path
#ifndef PXL_T2_UNROLL
#define PXL_T2_UNROLL 2
#endif
#pragma unroll PXL_T2_UNROLL
    for (int q = 0; q < PXL_TH / 4; ++q) {
        const int r = w + 4 * q;
        const int64_t jr = tj0 + r;
        if (jr >= p.nyo) break;
        double x[2], y[2];
        int32_t i0[2], j0[2];
        double fx[2], fy[2];
        bool fast = true;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            coords(r, e, &x[e], &y[e]);
            split_cell(x[e], &i0[e], &fx[e]);
            split_cell(y[e], &j0[e], &fy[e]);
            // both columns i0, i0 + 1 inside the map (no seam, no edge), both rows inside, finite coordinates
            const bool ok = isfinite(x[e]) && isfinite(y[e]) && j0[e] >= 1 && j0[e] < p.ny && i0[e] >= 1 && i0[e] < p.nx;
            fast = fast && (ok || (e == 1 && !two));
        }
        if (!__all(fast)) { slow_rows |= 1u << q; continue; }
        const int64_t t = jr * p.nxo + i;
        const int64_t o0 = (int64_t)(j0[0] - 1) * p.nx + (i0[0] - 1);
        const int64_t o1 = two ? (int64_t)(j0[1] - 1) * p.nx + (i0[1] - 1) : o0;
        for (int c = 0; c < p.nc; ++c) {
            const double* pl = p.src + (int64_t)c * p.nx * p.ny;
            const PxlPair t0 = *reinterpret_cast<const PxlPair*>(pl + o0), b0 = *reinterpret_cast<const PxlPair*>(pl + o0 + p.nx);
            const PxlPair t1 = *reinterpret_cast<const PxlPair*>(pl + o1), b1 = *reinterpret_cast<const PxlPair*>(pl + o1 + p.nx);
            const double v0 = (1 - fy[0]) * ((1 - fx[0]) * t0.a + fx[0] * t0.b) + fy[0] * ((1 - fx[0]) * b0.a + fx[0] * b0.b);
            const double v1 = (1 - fy[1]) * ((1 - fx[1]) * t1.a + fx[1] * t1.b) + fy[1] * ((1 - fx[1]) * b1.a + fx[1] * b1.b);
            double* o = p.dst + (int64_t)c * total + t;
            if (vec_store) {
                typedef double d2v __attribute__((ext_vector_type(2)));
                *reinterpret_cast<d2v*>(o) = d2v{v0, v1};
            } else { o[0] = v0; if (two) o[1] = v1; }
        }
    }
    // the rare rows (the seam of a periodic source, map edges, non-finite coordinates): per pixel, the path of round 3's kernel
#pragma unroll 1
    while (slow_rows) {
        const int q = __builtin_ctz(slow_rows);
        slow_rows &= slow_rows - 1;
        const int r = w + 4 * q;
        const int64_t t = (tj0 + r) * p.nxo + i;
        double xs[2], ys[2];
        coords(r, 0, &xs[0], &ys[0]);             // (compile-time column index: a run-time one puts colx / coly into scratch)
        coords(r, 1, &xs[1], &ys[1]);
#pragma unroll 1
        for (int e = 0; e < (two ? 2 : 1); ++e) generic_store(p, t + e, e ? xs[1] : xs[0], e ? ys[1] : ys[0], true);
    }
}

