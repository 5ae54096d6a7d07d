// generic_ring_kernel.h -- NOT COMPILED INTO THE LIBRARY.  The LDS-ring form of the CAR<->Gnomonic pixel kernel that VERDICT r03 item 4
// asked for (the structure that won R1: a wave marching down its tile with the needed source rows DMA'd into an LDS ring, taps out of
// LDS), as it was built and measured in round 4 (it drops into pixell.jl_amd/csrc/pxl_sample.h after k_reproject_generic_tiled3 and
// is launched with grid (2 gx, gy), 64 threads, PXL_GR_NS * PXL_GR_SEG * 8 bytes of dynamic LDS).
//
// RESULT (MI355X, 16-patch mosaic of 4096^2 Gnomonic patches from the 0.5' full-sky map, tools/research/r04_15.sh):
//     k_reproject_generic_tiled3 (direct 16-byte taps, rows in groups of four)        1.327 ms
//     this kernel, 16-slot ring, 4 rows ahead                                          2.419 ms
//     2 rows ahead                                                                     2.339 ms
//     32-slot ring (32 KB of LDS per wave), 4 / 8 rows ahead                           3.708 / 3.667 ms
// bit-identical output in every case (same checksum; tools/fuzz_parity.py --only generic: 6 608 cases, 0 failures).
//
// WHY IT LOSES.  R1's wave moves 4 KiB of stores and 2-3 KiB of loads per output row; its ~2 us per row of wave-uniform bookkeeping and
// dependent waits (DMA landed -> LDS read -> blend -> store) is amortised over those bytes.  A rotated tile cannot be wide: the ring must
// hold every source row an output row touches, 2 + width x sin(rotation), so 64 columns is what 16 KB per wave affords at the 8-11
// degrees the corners of a 34-degree patch reach -- and a 64-pixel row is 512 B of output: the same per-row chain for an eighth of the
// bytes.  8 192 half-tiles x 32 rows x ~1 us over ~2 300 resident waves is the 114 us per patch measured.  The direct form hides its
// latency with four rows of loads in flight and 4 waves per SIMD instead.  (Round 2's bounding-box staging of a whole tile, +-5 %, and its
// persistent double-buffered variant, 1.7x slower, failed for the same reason: LDS capacity against the rotation span.)

// Round 4, the R1 structure for the non-separable case: ONE WAVE PER 64 x 32 HALF-TILE, the source rows
// it needs DMA'd into an LDS ring, taps out of LDS.  Why: k_reproject_generic_tiled3 keeps the texture addresser busy 75 % of its
// launch -- neighbouring lanes' 16-byte taps overlap by 8 bytes and every one of them is a separate L1 access (32 per tap
// instruction, profiles/r04_tan_mosaic_counters.txt).  A source row segment brought in by ONE `global_load_lds_dwordx4` (1 KiB,
// 16-byte aligned, coalesced) costs the addresser a quarter of that, and the taps become LDS reads.
//   * half-tile = columns [64 h, 64 h + 64) of a lattice tile: its coordinates are inside the hull of lattice nodes 3 h .. 3 h + 3
//     (columns 0, 21.2, 42.3, 63.5 | 63.5, 84.7, 105.8, 127), so the nodes bound the source window: columns
//     [cbase0, cbase0 + 128), rows [jlo, jhi]
//   * ring: 16 slots of 128 doubles (slot = row & 15); rows are requested in order of the march (t = +-row as in
//     k_reproject_dma), PXL_GR_AHEAD rows beyond what the next output row reads, never onto a live slot; exact vmcnt waits with
//     per-slot marks
//   * the rows an output row reads: lanes 0 and the last active one bound them when y is monotone along the row; one row of margin
//     either side covers an extremum inside (curvature is 1e-3 pixel); every lane is then CHECKED against the window (one vote) and
//     a row that fails is done by direct taps -- like every half-tile that is not eligible (window too wide / tall for the ring,
//     the seam of a periodic source, map edges, odd nx) and every pixel whose cell is not interior
//   * the blend is generic_store's interior expression: bit-identical to the other forms
#ifndef PXL_GR_AHEAD
#define PXL_GR_AHEAD 4
#endif
#ifndef PXL_GR_NS
#define PXL_GR_NS 16
#endif
#define PXL_GR_SEG 128
__global__ __launch_bounds__(64) void k_reproject_generic_ring(GenericParams p, const double2* __restrict__ lat,
                                                               const int32_t* __restrict__ flag, int64_t gx) {
    extern __shared__ __attribute__((aligned(16))) double gr_ring[];      // PXL_GR_NS * PXL_GR_SEG doubles
    const int lane = threadIdx.x;
    const int half = (int)(blockIdx.x & 1);
    const int64_t bx = blockIdx.x >> 1, by = blockIdx.y;
    const int64_t tile = by * gx + bx;
    const int64_t ti0 = bx * PXL_TW, tj0 = by * PXL_TH;
    if (tile == 0 && half == 0 && lane == 0 && p.exact_tiles_next) *p.exact_tiles_next = 0u;
    if (ti0 + 64 * half >= p.nxo) return;
    if (flag[tile]) return;                      // k_reproject_generic_exact_tiles does this tile
    const int cx = 64 * half + lane;
    const int64_t i = ti0 + cx;
    const bool active = i < p.nxo;
    const int last_lane = (int)((p.nxo - (ti0 + 64 * half)) < 64 ? (p.nxo - (ti0 + 64 * half)) - 1 : 63);
    const int nrows = (int)((p.nyo - tj0) < PXL_TH ? (p.nyo - tj0) : PXL_TH);
    const double2* L = lat + tile * (PXL_TNX * PXL_TNY);
    double colx[PXL_TNY], coly[PXL_TNY];
    {
        double wx[PXL_TNX];
#pragma unroll
        for (int a = 0; a < PXL_TNX; ++a) wx[a] = c_tile_weights.wx[cx][a];
#pragma unroll
        for (int b = 0; b < PXL_TNY; ++b) {
            double sx = 0.0, sy = 0.0;
#pragma unroll
            for (int a = 0; a < PXL_TNX; ++a) { const double2 v = L[b * PXL_TNX + a]; sx = __builtin_fma(wx[a], v.x, sx); sy = __builtin_fma(wx[a], v.y, sy); }
            colx[b] = sx; coly[b] = sy;
        }
    }
    // ---- the source window of this half-tile from its lattice nodes (wave-uniform)
    double xmn = 1e300, xmx = -1e300, ymn = 1e300, ymx = -1e300, span = 0.0;
    bool fin = true;
#pragma unroll
    for (int b = 0; b < PXL_TNY; ++b) {
        double rmn = 1e300, rmx = -1e300;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const double2 v = L[b * PXL_TNX + 3 * half + a];
            fin = fin && v.x == v.x && v.y == v.y;
            xmn = fmin(xmn, v.x); xmx = fmax(xmx, v.x); rmn = fmin(rmn, v.y); rmx = fmax(rmx, v.y);
        }
        ymn = fmin(ymn, rmn); ymx = fmax(ymx, rmx); span = fmax(span, rmx - rmn);
    }
    const bool dypos = L[(PXL_TNY - 1) * PXL_TNX + 3 * half].y >= L[3 * half].y;
    bool eligible = fin && fabs(xmn) < 1e9 && fabs(xmx) < 1e9 && fabs(ymn) < 1e9 && fabs(ymx) < 1e9;
    int cbase0 = 0;
    if (eligible) {
        cbase0 = ((int)floor(xmn) - 4) & ~1;                                       // 0-based source column of ring element 0 (even)
        const int kmax = (int)floor(xmx) + 3 - cbase0;                            // highest element a tap may read (+ margin)
        const int jlo = (int)floor(ymn) - 2, jhi = (int)floor(ymx) + 3;           // source rows the half-tile may touch (1-based)
        eligible = cbase0 >= 0 && (int64_t)cbase0 + PXL_GR_SEG <= p.nx && kmax < PXL_GR_SEG && jlo >= 1 && jhi <= p.ny &&
                   (int)ceil(span) + 5 + PXL_GR_AHEAD <= PXL_GR_NS && (p.nx & 1) == 0 && (((uintptr_t)p.src & 15) == 0);
    }
    const int64_t total = p.nxo * p.nyo;
    const uint32_t nxm1 = (uint32_t)(p.nx - 1), nym1 = (uint32_t)(p.ny - 1);
    const uint32_t lds_base = (uint32_t)(uintptr_t)gr_ring;
    struct __attribute__((packed, aligned(8))) Pair { double a, b; };
    struct Row { double fx, fy; int32_t i0, j0; };
    auto row_coords = [&](int r) {
        double sx = 0.0, sy = 0.0;
#pragma unroll
        for (int b = 0; b < PXL_TNY; ++b) { const double wy = c_tile_weights.wy[r][b]; sx = __builtin_fma(wy, colx[b], sx); sy = __builtin_fma(wy, coly[b], sy); }
        const double flx = floor(sx), fly = floor(sy);
        return Row{sx - flx, sy - fly, (int32_t)flx, (int32_t)fly};                 // saturating conversions; NaN -> 0
    };
    for (int c = 0; c < p.nc; ++c) {
        const double* pl = p.src + (int64_t)c * p.nx * p.ny;
        double* dpl = p.dst + (int64_t)c * total + tj0 * p.nxo + i;
        // ring state (wave-uniform)
        int treq = 0, vm_total = 0, marks = 0;
        bool started = false;
        auto issue_next = [&]() {
            ++treq;
            int j = dypos ? treq : -treq;                                           // 1-based source row of t
            j = j < 1 ? 1 : (j > (int)p.ny ? (int)p.ny : j);                          // (rows requested ahead of the map's end: any readable row)
            const double* rowp = pl + (int64_t)(j - 1) * p.nx + cbase0;
            glds16_saddr((const void*)rowp, (uint32_t)(16 * lane), lds_base + (uint32_t)(treq & (PXL_GR_NS - 1)) * (PXL_GR_SEG * 8u));
            vm_total += 1;
            marks = (lane == (treq & (PXL_GR_NS - 1))) ? vm_total : marks;
        };
        Row cur = row_coords(0);
#pragma unroll 1
        for (int r = 0; r < nrows; ++r) {
            const Row nxt = (r + 1 < nrows) ? row_coords(r + 1) : cur;
            const bool lean = (uint32_t)(cur.i0 - 1) < nxm1 && (uint32_t)(cur.j0 - 1) < nym1;     // all four taps inside, no seam
            bool done = false;
            if (eligible) {
                // rows in t space (t runs with the march): the output row reads t0 and t0 + 1 per lane
                const int t0 = dypos ? cur.j0 : -(cur.j0 + 1), t0n = dypos ? nxt.j0 : -(nxt.j0 + 1);
                const int ta = __builtin_amdgcn_readlane(t0, 0), tb = __builtin_amdgcn_readlane(t0, last_lane);
                const int tna = __builtin_amdgcn_readlane(t0n, 0), tnb = __builtin_amdgcn_readlane(t0n, last_lane);
                const int tmin = (ta < tb ? ta : tb) - 1, tmax = (ta > tb ? ta : tb) + 1;      // rows tmin .. tmax + 1 cover this output row
                const int tnmax = (tna > tnb ? tna : tnb) + 1;
                if (!started || treq < tmin - 1) { treq = tmin - 1; started = true; }             // (rows below tmin are dead: skip them)
                const int k = cur.i0 - 1 - cbase0;
                const bool inwin = !active || (lean && t0 >= tmin && t0 <= tmax && (uint32_t)k < (uint32_t)(PXL_GR_SEG - 1));
                const bool fits = tmax + 1 - tmin < PXL_GR_NS && tmin > treq - PXL_GR_NS;       // the rows fit the ring and none was overwritten
                if (fits && __all(inwin)) {
                    int want = (tmax > tnmax ? tmax : tnmax) + 1 + PXL_GR_AHEAD;
                    const int lim = tmin + PXL_GR_NS - 1;                                       // never onto a live slot
                    if (want > lim) want = lim;
                    while (treq < want) issue_next();
                    wait_vm_upto(vm_total - __builtin_amdgcn_readlane(marks, (tmax + 1) & (PXL_GR_NS - 1)));
                    const int kk = active ? k : 0;                                               // (inactive lanes read element 0)
                    const double* Ra = gr_ring + (t0 & (PXL_GR_NS - 1)) * PXL_GR_SEG + kk;         // the older row in t space
                    const double* Rb = gr_ring + ((t0 + 1) & (PXL_GR_NS - 1)) * PXL_GR_SEG + kk;
                    const double a0 = Ra[0], a1 = Ra[1], b0 = Rb[0], b1 = Rb[1];
                    const double ha = (1 - cur.fx) * a0 + cur.fx * a1, hb = (1 - cur.fx) * b0 + cur.fx * b1;
                    const double top = dypos ? ha : hb, bot = dypos ? hb : ha;                  // row j0 / row j0 + 1
                    const double v = (1 - cur.fy) * top + cur.fy * bot;
                    if (active) dpl[(int64_t)r * p.nxo] = v;
                    vm_total += 1;                                                              // the store (lane 0 is always active)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                       // the reads are done before a later DMA may reuse a slot
                    done = true;
                }
            }
            if (!done) {
                if (__all(!active || lean)) {
                    if (active) {
                        const double* q = pl + (int64_t)(cur.j0 - 1) * p.nx + (cur.i0 - 1);
                        const Pair tp = *reinterpret_cast<const Pair*>(q), bt = *reinterpret_cast<const Pair*>(q + p.nx);
                        const double top = (1 - cur.fx) * tp.a + cur.fx * tp.b;
                        const double bot = (1 - cur.fx) * bt.a + cur.fx * bt.b;
                        dpl[(int64_t)r * p.nxo] = (1 - cur.fy) * top + cur.fy * bot;
                    }
                } else if (active) {
                    double sx = 0.0, sy = 0.0;
#pragma unroll
                    for (int b = 0; b < PXL_TNY; ++b) { const double wy = c_tile_weights.wy[r][b]; sx = __builtin_fma(wy, colx[b], sx); sy = __builtin_fma(wy, coly[b], sy); }
                    // generic_store writes every component; call it for the first one only
                    if (c == 0) generic_store(p, (tj0 + r) * p.nxo + i, sx, sy, true);
                }
                // direct loads and stores of unknown number: the counts above are no longer exact -- drain and restart the ring
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                vm_total = 0; marks = 0; started = false;
            }
            cur = nxt;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // no DMA may land in LDS after this wave has moved on (or ended)
    }
}

