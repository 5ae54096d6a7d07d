// pxl_reproject_dma.h -- the LDS-DMA reprojection kernel (fast path of R1); included by pxl_kernels.hip.
//
// ONE WAVEFRONT PER OUTPUT TILE.  A tile is TW = 128*PAIRS output columns x rh output rows of one
// component plane; the wave marches down the tile's rows.  Source rows go HBM -> LDS directly with
// `global_load_lds_dwordx4` (16 B per lane, 1 KiB per instruction, no VGPR staging) into a ring of
// `ns` LDS slots, several rows ahead of the row being interpolated: the ring is the in-flight buffer
// that hides HBM latency, since each wave's loads are otherwise a dependent chain.
//
// The kernel requires the tile's source-row sequence to be monotone (it is, for an affine CAR -> CAR
// row map, except across a y-rewind jump); rows are then requested strictly in order, one ring slot per
// row (slot = t & (ns-1), t = +-row), and all bookkeeping is wave-uniform scalar arithmetic:
//   treq        highest t requested so far
//   need        t0 + 1, the farther of the two rows the current output row reads
//   vm_total    vector-memory instructions this wave has issued (DMAs + its own 16-byte stores)
//   marks       lane s: vm_total right after the DMAs of the row in ring slot s
//   s_waitcnt vmcnt(vm_total - marks[need])   -- exactly what was issued after row `need`'s DMAs: vmcnt retires in
// issue order, stores included, so a wait that ignored the stores (round 2: vmcnt((treq - need) * NCH)) also waited
// for the acknowledgement of the wave's own recent stores and cut the effective prefetch distance in half.
// Each source row is read from LDS ONCE per tile: its horizontal interpolant at the lane's columns is kept in
// registers and serves as `top` and as `bot` of every output row that touches it (bit-identical: same operations).
// Tiles that are not monotone, or whose columns do not fit the slot, take the direct-tap fallback.
//
// RA seam: slot element k holds source column (cbase0 + k) mod nx, resolved in the per-lane source
// address of the DMA, so the interpolation never sees the seam.  Columns outside a non-periodic map and
// rows outside the map/window read a 16-byte zero page instead (every lane always issues, LDS needs no
// zero fill, and the DMA count stays exact).
#pragma once

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// wait until at most k vector-memory instructions of this wave are outstanding (k rounded DOWN to an immediate the
// switch carries: a stricter wait is always safe).  vmcnt is a 6-bit counter on gfx9 and retires in issue order,
// loads and stores alike.
__device__ __forceinline__ void wait_vm_upto(int k) {
    // a hand-written comparison tree (the compiler turns a dense switch over immediates into a chain of flag tests)
    if (k < 8) {
        if (k < 4) {
            if (k < 2) { if (k < 1) wait_vm<0>(); else wait_vm<1>(); }
            else       { if (k < 3) wait_vm<2>(); else wait_vm<3>(); }
        } else {
            if (k < 6) { if (k < 5) wait_vm<4>(); else wait_vm<5>(); }
            else       { if (k < 7) wait_vm<6>(); else wait_vm<7>(); }
        }
    } else if (k < 16) {
        if (k < 12) {
            if (k < 10) { if (k < 9) wait_vm<8>(); else wait_vm<9>(); }
            else        { if (k < 11) wait_vm<10>(); else wait_vm<11>(); }
        } else {
            if (k < 14) { if (k < 13) wait_vm<12>(); else wait_vm<13>(); }
            else        { if (k < 15) wait_vm<14>(); else wait_vm<15>(); }
        }
    } else {
        if (k < 24) { if (k < 20) wait_vm<16>(); else wait_vm<20>(); }
        else        { if (k < 32) wait_vm<24>(); else wait_vm<32>(); }
    }
}

// One 1-KiB LDS-DMA: lane l copies 16 B from (sbase + voff[l]) to LDS[lds_byte_addr + 16*l].
// M0 carries the LDS destination and is written in the same statement that uses it.
__device__ __forceinline__ void glds16_saddr(const void* sbase, uint32_t voff, uint32_t lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %0"
                 :: "s"(sbase), "v"(voff), "s"(lds_byte_addr) : "memory");
}
__device__ __forceinline__ void glds16_vaddr(const void* gsrc, uint32_t lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 :: "v"(gsrc), "s"(lds_byte_addr) : "memory");
}

// T = storage type of the maps (double or float); EPL = elements per lane per 16-byte access.
template <typename T, int PAIRS, int NCH>
__global__ __launch_bounds__(64) void k_reproject_dma(ReprojParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];   // ns * seg elements of T
    T* const lds = reinterpret_cast<T*>(lds_raw);
    const int lane = threadIdx.x;
    constexpr int EPL = 16 / (int)sizeof(T);       // 2 doubles or 4 floats per lane per access
    constexpr int CW = 64 * EPL;                   // elements per wave access (1 KiB)
    constexpr int TW = CW * PAIRS;                 // output columns per tile

    // XCD-aware decode: blocks b and b+8 share an XCD; give each XCD a contiguous run of tiles so that
    // RA-neighbouring tiles (shared 128-B lines at the edges, same source rows) meet in one L2.
    const int64_t b = blockIdx.x;
    const int64_t t = (p.flags & 4) ? b : xcd_tile(b, p.xchunk, (p.flags & 8) ? 1 : ((p.flags & 16) ? 2 : 0));
    if (t >= p.ntiles) return;
    // RA-fastest tile order (DEC-fastest, non-temporal loads and non-temporal stores were all measured:
    // each within 1 % of this; profiles/r01_tuning_sweeps.log)
    const int tx = (int)(t % p.ntx);
    const int64_t trest = t / p.ntx;
    const int ty = (int)(trest % p.nty);
    const int c = (int)(trest / p.nty);

    const T* splane = (const T*)p.src + (int64_t)c * p.nx * p.src_nrows;
    T* dplane = (T*)p.dst + (int64_t)c * p.nxo * p.dst_nrows;

    const int64_t c0 = (int64_t)tx * TW;
    const int64_t clast = (c0 + TW < p.nxo ? c0 + TW : p.nxo) - 1;
    const int64_t rb = p.r0 + (int64_t)ty * p.rh;
    const int64_t re = (rb + p.rh < p.r0 + p.nr) ? rb + p.rh : p.r0 + p.nr;
    const int nrows = (int)(re - rb);

    // ---- per-lane column setup (constant over the tile)
    const int64_t a = p.dxpos ? p.xi0[c0] : p.xi0[clast];      // 1-based source cell at the tile's low end
    const int64_t ua = a - 1;
    const int64_t cbase0 = ua - (((ua % EPL) + EPL) % EPL);    // 0-based source column at slot index 0 (16-B aligned)
    int dloc[PAIRS][EPL];
    double fx[PAIRS][EPL];
    bool act[PAIRS][EPL];
    bool fits = true;
#pragma unroll
    for (int q = 0; q < PAIRS; ++q) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            int64_t col = c0 + q * CW + EPL * lane + e;
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

    // ---- row table of this tile: lane l holds output row rb + l; broadcast with v_readlane
    int my_j0 = 0;
    double my_fy = 0.0;
    if (lane < nrows) { my_j0 = p.yj0[p.dst_row0 + rb + lane]; my_fy = p.yfy[p.dst_row0 + rb + lane]; }
    const int dir = p.dypos ? 1 : -1;
    const int my_t0 = p.dypos ? my_j0 : -(my_j0 + 1);          // rows needed: t0, t0+1 in t = dir*row space
    // vertical weights of the older (t0) and the newer (t0 + 1) row in t space: (1 - fy, fy) when t runs with the
    // source rows, swapped when it runs against them
    const double my_wa = p.dypos ? 1 - my_fy : my_fy;
    const double my_wb = p.dypos ? my_fy : 1 - my_fy;
    {
        int nxt = __shfl_down(my_t0, 1, 64);
        bool mono = (lane + 1 >= nrows) || (nxt >= my_t0);
        if (!mono) fits = false;
    }

    if (!__all(fits)) {
        // wave-uniform fallback: direct taps (rewind discontinuity inside the tile)
        SrcViewT<T> m{splane, p.nx, p.ny, p.src_row0, p.src_nrows, p.periodic};
        for (int64_t r = rb; r < re; ++r) {
            int64_t j0 = p.yj0[p.dst_row0 + r];
            double fy = p.yfy[p.dst_row0 + r];
#pragma unroll
            for (int q = 0; q < PAIRS; ++q)
#pragma unroll
                for (int e = 0; e < EPL; ++e) {
                    int64_t col = c0 + q * CW + EPL * lane + e;
                    if (act[q][e]) dplane[r * p.nxo + col] = (T)bilerp_cells(m, p.xi0[col], fx[q][e], j0, fy);
                }
        }
        return;
    }

    // ---- per-lane source byte offsets of each 1-KiB chunk within a source row (constant over the tile);
    //      0xFFFFFFFF marks a column outside a non-periodic map
    uint32_t voff[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        int64_t u = cbase0 + ch * CW + EPL * lane;
        bool ok = true;
        if (p.periodic) { u %= p.nx; if (u < 0) u += p.nx; }
        else ok = (u >= 0) && (u < p.nx);
        voff[ch] = ok ? (uint32_t)(u * (int64_t)sizeof(T)) : 0xFFFFFFFFu;
    }

    const uint32_t lds_base = (uint32_t)(uintptr_t)lds;
    const uint32_t slot_bytes = (uint32_t)p.seg * (uint32_t)sizeof(T);
    const int ns_mask = p.ns - 1;

    // ---- request bookkeeping (all wave-uniform).  Rows are requested strictly in t order, so the row
    // pointer advances by a constant stride and validity is a range test in t space.
    int tv_lo = 1, tv_hi = 0;                                   // valid t range (empty by default)
    {
        const int64_t jlo = (p.src_row0 + 1 > 1) ? p.src_row0 + 1 : 1;                    // first resident in-map row
        const int64_t jhi = (p.src_row0 + p.src_nrows < p.ny) ? p.src_row0 + p.src_nrows : p.ny;
        if (jhi >= jlo && !(p.flags & 1)) {
            if (p.dypos) { tv_lo = (int)jlo; tv_hi = (int)jhi; } else { tv_lo = -(int)jhi; tv_hi = -(int)jlo; }
        }
    }
    const int64_t row_step = (int64_t)dir * p.nx * (int64_t)sizeof(T);   // bytes from row t to row t + 1
    int treq = __builtin_amdgcn_readlane(my_t0, 0) - 1;         // nothing requested yet
    const char* next_row = (const char*)splane + ((int64_t)dir * (treq + 1) - 1 - p.src_row0) * p.nx * (int64_t)sizeof(T);
    const bool tail_full = (p.seg == NCH * CW);
    const bool in_tail = ((NCH - 1) * CW + EPL * lane) < p.seg;  // lanes of the last chunk inside the slot
    const uint32_t zero_off = 0u;

    // Exact waits: vmcnt retires in issue order (loads, LDS-DMAs and stores alike), so "row t has landed" is "at most
    // as many VM instructions outstanding as were issued AFTER row t's last DMA".  vm_total counts what this wave has
    // issued (never more than it really has: an under-count only makes a wait stricter); lane s of `marks` remembers
    // vm_total just after the DMAs of the row in ring slot s.  Stores are counted only in tiles where every lane
    // stores with 16-byte stores (then each of the PAIRS store instructions of a row is certain to issue).
    int vm_total = 0;
    int marks = 0;
    auto issue_next = [&]() {           // request source row t = treq + 1 into slot t & ns_mask
        ++treq;
        const bool valid = (treq >= tv_lo) && (treq <= tv_hi);
        const uint32_t slot_addr = lds_base + (uint32_t)(treq & ns_mask) * slot_bytes;
        if (p.periodic) {
            // every lane has a valid column: SGPR row base + 32-bit per-lane byte offset
            const void* base = valid ? (const void*)next_row : (const void*)p.zero_page;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const uint32_t off = valid ? voff[ch] : zero_off;
                if (ch < NCH - 1 || tail_full) glds16_saddr(base, off, slot_addr + ch * 1024u);
                else if (in_tail) glds16_saddr(base, off, slot_addr + ch * 1024u);        // lane 0 is always in
            }
        } else {
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const void* g = (valid && voff[ch] != 0xFFFFFFFFu) ? (const void*)(next_row + voff[ch])
                                                                   : (const void*)p.zero_page;
                if (ch < NCH - 1 || tail_full) glds16_vaddr(g, slot_addr + ch * 1024u);
                else if (in_tail) glds16_vaddr(g, slot_addr + ch * 1024u);
            }
        }
        next_row += row_step;
        vm_total += NCH;
        marks = (lane == (treq & ns_mask)) ? vm_total : marks;
    };

    const bool vec_store = ((p.nxo % EPL) == 0) && (((uintptr_t)p.dst & 15) == 0);
    T* orow = dplane + rb * p.nxo + c0 + EPL * lane;           // this lane's first output group in row rb

    if (p.flags & 64) {
        // diagnostics: the tile's stores alone (no DMA, no LDS, no arithmetic) -- the write ceiling of this
        // tile shape and order
        for (int rr = 0; rr < nrows; ++rr) {
#pragma unroll
            for (int q = 0; q < PAIRS; ++q) {
                T* o = orow + q * CW;
                alignas(16) T v[EPL];
#pragma unroll
                for (int e = 0; e < EPL; ++e) v[e] = (T)fx[q][e];
                if (act[q][0] && vec_store) *reinterpret_cast<uint4*>(o) = *reinterpret_cast<const uint4*>(v);
            }
            orow += p.nxo;
        }
        return;
    }
    // ---- the row loop.  The horizontal interpolant of a source row at this lane's columns,
    //          h(t) = (1 - fx) * row_t[d] + fx * row_t[d + 1],
    // is the `top` of every output row whose upper source row is t and the `bot` of every output row whose lower one
    // is -- the same operations on the same operands either way -- so it is formed ONCE per source row and kept in
    // registers: hA = h(th), hB = h(th + 1) in t space.  An output row then costs one vertical blend
    // (1 - fy) * top + fy * bot; the LDS is read once per source row and tile instead of twice per output row
    // (2x refinement: a quarter of the LDS reads and half the FP64 work of the tap-per-output form, same bits).
    double hA[PAIRS][EPL], hB[PAIRS][EPL], wx[PAIRS][EPL];
#pragma unroll
    for (int q = 0; q < PAIRS; ++q)
#pragma unroll
        for (int e = 0; e < EPL; ++e) { wx[q][e] = 1 - fx[q][e]; hA[q][e] = 0.0; hB[q][e] = 0.0; }
    auto hrow = [&](int tr, double (&h)[PAIRS][EPL]) {
        const T* R = lds + (tr & ns_mask) * p.seg;
#pragma unroll
        for (int q = 0; q < PAIRS; ++q)
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                const int d = dloc[q][e];
                h[q][e] = wx[q][e] * (double)R[d] + fx[q][e] * (double)R[d + 1];
            }
    };
    // FAST: a tile with every lane inside the map, 16-byte stores and no diagnostics flag -- the stores are unconditional (no
    // exec masking, no flag tests: a third of the loop's scalar instructions), and each of them is certain to issue, so they are
    // counted for the exact waits.  Edge tiles and the diagnostic launches take the general form.
    auto row_loop = [&](auto fast_tag, auto nt_tag) {
    constexpr bool FAST = decltype(fast_tag)::value;
    constexpr bool NT = decltype(nt_tag)::value;
    int th = INT32_MIN / 2;                                     // no row interpolated yet
    for (int rr = 0; rr < nrows; ++rr) {
        const int t0 = __builtin_amdgcn_readlane(my_t0, rr);
        // top up the ring: rows up to the farther row of output row rr + pf, but never onto a live slot
        {
            const int ra = (rr + p.pf < nrows) ? rr + p.pf : nrows - 1;
            int tmax = __builtin_amdgcn_readlane(my_t0, ra) + 1;
            const int tlim = t0 + p.ns - 1;
            if (tmax > tlim) tmax = tlim;
            while (treq < tmax) issue_next();
        }
        if (t0 != th) {
            // rows t0 and t0 + 1 are needed; t0 + 1 was requested last of the two
            wait_vm_upto(vm_total - __builtin_amdgcn_readlane(marks, (t0 + 1) & ns_mask));
            if (t0 == th + 1) {
#pragma unroll
                for (int q = 0; q < PAIRS; ++q)
#pragma unroll
                    for (int e = 0; e < EPL; ++e) hA[q][e] = hB[q][e];
            } else {
                hrow(t0, hA);
            }
            hrow(t0 + 1, hB);
            th = t0;
            // the LDS reads above have returned (their values were consumed); make that explicit before a later DMA
            // may overwrite their slots
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        // vertical blend.  t runs with the source row when dypos (top = hA) and against it otherwise (top = hB);
        // (1 - fy) * top + fy * bot is formed as wa * hA + wb * hB with the weights swapped instead of the rows: the
        // two products are the same and IEEE addition commutes, so the bits are those of the oracle's order.
        const double wa = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(my_wa), rr),
                                           __builtin_amdgcn_readlane(__double2loint(my_wa), rr));
        const double wb = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(my_wb), rr),
                                           __builtin_amdgcn_readlane(__double2loint(my_wb), rr));
#pragma unroll
        for (int q = 0; q < PAIRS; ++q) {
            alignas(16) T v[EPL];
#pragma unroll
            for (int e = 0; e < EPL; ++e) v[e] = (T)(wa * hA[q][e] + wb * hB[q][e]);
            T* o = orow + q * CW;
            if (FAST) {
                typedef unsigned int u4v __attribute__((ext_vector_type(4)));
                if (NT) __builtin_nontemporal_store(*reinterpret_cast<const u4v*>(v), reinterpret_cast<u4v*>(o));
                else *reinterpret_cast<uint4*>(o) = *reinterpret_cast<const uint4*>(v);
            }
            else if (p.flags & 2) { if (v[0] == (T)1.2345e30) o[0] = v[1]; }       // diagnostics: keep v live, never store
            else if (vec_store) { if (act[q][0]) *reinterpret_cast<uint4*>(o) = *reinterpret_cast<const uint4*>(v); }
            else {
#pragma unroll
                for (int e = 0; e < EPL; ++e) if (act[q][e]) o[e] = v[e];
            }
        }
        if (FAST) vm_total += PAIRS;
        orow += p.nxo;
    }
    };
    // Non-temporal stores (p.nt, chosen at plan creation): the output then streams through the L2 without evicting the column
    // tables that every tile re-reads -- with ordinary stores those re-reads miss the L2 and show up as 3-6 % of extra FETCH_SIZE
    // (1.45 GB per launch of the IQU map; profiles/r03_fetch_by_variant_cfg3.txt) -- worth 0.7 % on the same-resolution IQU map;
    // the 2x refinement is 1.7 % slower with them and keeps ordinary stores.
    if (vec_store && (c0 + TW <= p.nxo) && p.flags == 0) {
        if (p.nt) row_loop(std::true_type{}, std::true_type{});
        else row_loop(std::true_type{}, std::false_type{});
    } else row_loop(std::false_type{}, std::false_type{});
}

template <typename T>
static int launch_reproject_dma_t(int pairs, int nch, dim3 grid, size_t lds_bytes, hipStream_t st, const ReprojParams& p) {
#define PXL_DMA_CASE(P, N) \
    if (pairs == P && nch == N) { hipLaunchKernelGGL((k_reproject_dma<T, P, N>), grid, dim3(64), lds_bytes, st, p); return check_launch("k_reproject_dma"); }
    PXL_DMA_CASE(1, 1) PXL_DMA_CASE(1, 2) PXL_DMA_CASE(1, 3) PXL_DMA_CASE(1, 4) PXL_DMA_CASE(1, 5)
    PXL_DMA_CASE(2, 1) PXL_DMA_CASE(2, 2) PXL_DMA_CASE(2, 3) PXL_DMA_CASE(2, 4) PXL_DMA_CASE(2, 5)
    PXL_DMA_CASE(4, 1) PXL_DMA_CASE(4, 2) PXL_DMA_CASE(4, 3) PXL_DMA_CASE(4, 4) PXL_DMA_CASE(4, 5)
#undef PXL_DMA_CASE
    return fail(PXL_EINVAL, "reproject: no LDS-DMA kernel for pairs=%d nch=%d", pairs, nch);
}
