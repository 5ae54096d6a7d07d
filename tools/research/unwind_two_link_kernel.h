// tools/research/unwind_two_link_kernel.h -- NOT part of the library.  One-pass unwind! with TWO links per workgroup, so that the
// look-back of chunk A runs beside the arithmetic of chunk B (chunk A's values parked in LDS); built and measured in round 4 and
// dropped: the same bits (33 GPU parity tests, 783 fuzz cases), slower.
//
//   pix2sky!(safe=true), 1e8 points out of place, one MI355X (tools/research/r04_22.sh):
//     k_unwind_onepass (one link per 4096-point workgroup; ships)            0.857-0.882 ms
//     k_unwind_onepass2, U = 2 (A 2048 + B 1920 points, 42 VGPRs)            0.942-0.957 ms
//                        U = 3 (A 3072 + B 2880 points)                      0.865-0.908 ms
//                        U = 1                                               1.22 ms
//     for scale: k_unwind_onepass with the look-back compiled out (wrong answers, timing only): 0.67-0.69 ms
//
// Why it loses: twice the links per point (every look-back window reaches half as far), a third workgroup barrier, and the LDS
// round trip of chunk A; what it hides (one look-back per two chunks) is worth less than that.  One lesson kept: an UNROLLED
// scalar loop over the 16 per-wave sums in LDS puts 48-80 values in flight in as many registers (this kernel: 114 VGPRs, 45
// spilled at 64); gathering them with lanes 0..15 and a wave scan (uw2_gather) brought it to 42.
//
// ------------------------------------------------------------------------------------------------
// One pass, TWO links per workgroup (round 4, late): the look-back of k_unwind_onepass costs 0.175 of its 0.855 ms per 1e8 points
// (a timing build without it: 0.68 ms) because 15 waves wait at a barrier while wave 0 reads links across the chip.  Here a
// workgroup takes two consecutive chunks, A (all 16 waves, 2 x 64 points each) and B (waves 1..15, 2 x 64 points each), and a
// link for each.  After A's sums meet in LDS, wave 0 publishes A's aggregate and looks back for A WHILE waves 1..15 form B's
// rewound values and sums; the last of them to finish publishes B's aggregate (an LDS counter says who is last), so a successor
// never waits for this workgroup's look-back.  B needs no look-back: its carry is A's carry plus A's total.  Both chunks are
// applied after the second barrier.  Registers as before (four 16-byte points per lane).  Same arithmetic per element as
// k_unwind_onepass / k_unwind_apply, hence the same bits; deadlock freedom by the same argument (tickets; an aggregate is
// published before anything is waited for; every wait is for links with smaller ids).
// ------------------------------------------------------------------------------------------------
#ifndef PXL_UW2_U
#define PXL_UW2_U 2            // 64-point groups per wave and chunk
#endif
#define PXL_UW2_WAVES 16
#define PXL_UW2_SA (PXL_UW2_WAVES * 64 * PXL_UW2_U)              // points of chunk A
#define PXL_UW2_SB ((PXL_UW2_WAVES - 1) * 64 * PXL_UW2_U)        // points of chunk B
#define PXL_UW2_S (PXL_UW2_SA + PXL_UW2_SB)

// rewound values, packed increments and this wave's sums of U groups of 64 points starting at base
template <class SRC, int U>
__device__ inline void uw2_compute(const SRC& src, int lane, int64_t base, int64_t n, const typename SRC::raw_t* v, const double* mfirst,
                                   double (*m)[2], int* cc, int* T, unsigned int* nh) {
    constexpr int NROW = SRC::NROW;
    double mlast[2] = {mfirst[0], mfirst[1]};
    int sum[2] = {0, 0};
    bool nanl[2] = {false, false};
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t k = base + (int64_t)u * 64 + lane;
        double mp[2];
        int c[2] = {0, 0};
        uw_element(src, lane, k, k < n, v[u], mlast, m[u], mp, c);
        cc[u] = (c[0] + 1) | (NROW == 2 ? (c[1] + 1) << 16 : 0);
#pragma unroll
        for (int r = 0; r < NROW; ++r) {
            sum[r] += c[r];
            nanl[r] = nanl[r] || (k < n && m[u][r] != m[u][r]);
            mlast[r] = uw_lane63(m[u][r]);
        }
    }
    T[0] = uw_wave_total(sum[0]);
    T[1] = NROW == 2 ? uw_wave_total(sum[1]) : 0;
    *nh = (__ballot(nanl[0]) != 0ull ? 1u : 0u) | (NROW == 2 && __ballot(nanl[1]) != 0ull ? 2u : 0u);
}

// k_unwind_apply's arithmetic on values kept in registers; returns "an element disagreed with the reference's recurrence"
template <class SRC, int U>
__device__ inline bool uw2_apply(const SRC& src, typename SRC::raw_t* out, int lane, int64_t base, int64_t n, const double (*m)[2], const int* cc,
                                 const double* mfirst, int carry0, int carry1, unsigned int nan_before) {
    constexpr int NROW = SRC::NROW;
    const double P = src.period, rP = src.rperiod, ref = src.ref;
    int carry[2] = {carry0, carry1};
    bool pex[2] = {(nan_before & 1u) != 0, (nan_before & 2u) != 0};
    bool bad = false;
    double mlast[2] = {mfirst[0], mfirst[1]};
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t k = base + (int64_t)u * 64 + lane;
        const bool valid = k < n;
        const int s = uw_scan64(cc[u]);
        const int tot = __builtin_amdgcn_readlane(s, 63);
        double y[2] = {0.0, 0.0};
#pragma unroll
        for (int r = 0; r < NROW; ++r) {
            const double mp = uw_shr1_first(m[u][r], mlast[r]);
            mlast[r] = uw_lane63(m[u][r]);
            const int field = r == 0 ? (s & 0xffff) : (s >> 16);
            const int c = (r == 0 ? (cc[u] & 0xffff) : (cc[u] >> 16)) - 1;
            const int rr = carry[r] + field - (lane + 1);            // r_k
            carry[r] += (r == 0 ? (tot & 0xffff) : (tot >> 16)) - 64;
            const unsigned long long nanmask = __ballot(valid && m[u][r] != m[u][r]);
            const bool poisoned = pex[r] || (nanmask & ((2ull << lane) - 1ull)) != 0ull;
            pex[r] = pex[r] || nanmask != 0ull;
            if (!valid) continue;
            if (poisoned) { y[r] = __builtin_nan("") + ref; continue; }
            if (k > 0) {
                const double yprev = mp - (double)(rr - c) * P;       // y[k-1] as the reference forms it
                const double a = m[u][r] - yprev;
                const double qa = a * rP;
                if (!(fabs(qa - (double)rr) < 0.4999)) {
                    const double q = a / P;
                    if (!(rint(q) == (double)rr)) bad = true;
                }
            }
            y[r] = (m[u][r] - (double)(k > 0 ? rr : 0) * P) + ref;     // k = 0: m - 0 = m, bit for bit
        }
        if (valid) SRC::store(out, k, y);
    }
    return bad;
}

// the exclusive carry of link `id` (one wave): aggregates back to the nearest inclusive prefix, 64 links per round
template <int NROW>
__device__ inline void uw2_lookback(const UwLink* links, int64_t id, int lane, int* E, unsigned int* nan_before, bool* gave_up) {
    E[0] = E[1] = 0; *nan_before = 0; *gave_up = false;
    if (id <= 0) return;
    int64_t top = id - 1;
    unsigned int polls = 0;
    for (;;) {
        const int64_t j = top - lane;
        unsigned long long a = 0, p0 = 0, p1 = 0;
        if (j >= 0) {
            p0 = uw_peek(&links[j].pre0);
            p1 = NROW == 2 ? uw_peek(&links[j].pre1) : p0;
            a = uw_peek(&links[j].agg);
        }
        const bool hasP = j < 0 || (((unsigned)p0 & 1u) && ((unsigned)p1 & 1u));        // before the first chunk: a prefix of zero
        const bool hasA = j < 0 || ((unsigned)a & 1u);
        const unsigned long long pmask = __ballot(hasP), amask = __ballot(hasA);
        const int first = pmask ? __builtin_ctzll(pmask) : 64;                          // nearest link of this window with a prefix
        const unsigned long long need = first >= 64 ? ~0ull : ((1ull << first) - 1ull);
        if ((amask & need) != need) {                                                   // an aggregate in between is not there yet
            if (++polls > (1u << 22)) { *gave_up = true; return; }
            __builtin_amdgcn_s_sleep(2);
            continue;
        }
        int c0 = 0, c1 = 0;
        unsigned int nb = 0;
        if (j >= 0 && lane < first) {
            const unsigned int pa = (unsigned)(a >> 32);
            c0 = (int)(pa & 0x3fffu) - 8192; c1 = (int)((pa >> 14) & 0x3fffu) - 8192; nb = pa >> 28;
        } else if (j >= 0 && lane == first) {
            c0 = (int)(unsigned)(p0 >> 32); c1 = (int)(unsigned)(p1 >> 32);
            nb = (((unsigned)p0 >> 1) & 1u) | ((((unsigned)p1 >> 1) & 1u) << 1);
        }
        E[0] += uw_wave_total(c0);
        if (NROW == 2) E[1] += uw_wave_total(c1);
        if (__ballot(nb & 1u) != 0ull) *nan_before |= 1u;
        if (__ballot(nb & 2u) != 0ull) *nan_before |= 2u;
        if (first < 64) return;
        top -= 64;
        polls = 0;
    }
}

// The 16 per-wave sums of a chunk, gathered by lanes 0..15 and reduced in the wave (an unrolled scalar loop over LDS keeps 80
// values in flight in as many registers): totals, the part before wave `upto`, and the NaN flags likewise.  Waves below `from`
// hold no part of the chunk.
struct UwGather { int tot[2], before[2]; unsigned int nan_all, nan_before; };
__device__ inline UwGather uw2_gather(const int (*wsum)[PXL_UW2_WAVES], const int* wnan, int lane, int from, int upto) {
    const bool in = lane >= from && lane < PXL_UW2_WAVES;
    const int a0 = in ? wsum[0][lane] : 0, a1 = in ? wsum[1][lane] : 0;
    const unsigned int na = in ? (unsigned)wnan[lane] : 0u;
    const int s0 = uw_scan64(a0), s1 = uw_scan64(a1);
    UwGather g;
    g.tot[0] = __builtin_amdgcn_readlane(s0, 63); g.tot[1] = __builtin_amdgcn_readlane(s1, 63);
    g.before[0] = upto > 0 ? __builtin_amdgcn_readlane(s0, upto - 1) : 0;
    g.before[1] = upto > 0 ? __builtin_amdgcn_readlane(s1, upto - 1) : 0;
    const unsigned long long n0 = __ballot(na & 1u), n1 = __ballot(na & 2u);
    const unsigned long long below = upto >= 64 ? ~0ull : ((1ull << upto) - 1ull);
    g.nan_all = (n0 ? 1u : 0u) | (n1 ? 2u : 0u);
    g.nan_before = ((n0 & below) ? 1u : 0u) | ((n1 & below) ? 2u : 0u);
    return g;
}

template <class SRC>
__global__ __launch_bounds__(64 * PXL_UW2_WAVES) void k_unwind_onepass2(SRC src, typename SRC::raw_t* out, int64_t n, UwLink* __restrict__ links,
                                                                        unsigned int* __restrict__ ticket, int32_t* __restrict__ flag) {
    constexpr int U = PXL_UW2_U, NROW = SRC::NROW, NW = PXL_UW2_WAVES;
    __shared__ unsigned int id_s, doneB_s;
    __shared__ int wsumA_s[2][NW], wnanA_s[NW], wsumB_s[2][NW], wnanB_s[NW];
    __shared__ int excl_s[2];
    __shared__ unsigned int nanb_s, gaveup_s;
    // chunk A's rewound values and packed increments wait in LDS (40 KB of the CU's 160) while chunk B is formed, so that the
    // registers hold ONE chunk at a time (both in registers: 114 VGPRs, or 45 of them spilled at the 64 that two workgroups per CU allow)
    __shared__ double2 mA_s[U][64 * NW];
    __shared__ int ccA_s[U][64 * NW];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x == 0) { id_s = atomicAdd(ticket, 1u); gaveup_s = 0u; doneB_s = 0u; }
    __syncthreads();
    const int64_t t = (int64_t)(unsigned int)__builtin_amdgcn_readfirstlane((int)id_s);
    const int64_t baseA = t * PXL_UW2_S + (int64_t)wave * 64 * U;
    const int64_t baseB = t * PXL_UW2_S + PXL_UW2_SA + (int64_t)(wave - 1) * 64 * U;       // waves 1..15
    typename SRC::raw_t vA[U], vB[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t k = baseA + (int64_t)u * 64 + lane;
        vA[u] = (k < n) ? src.load(k) : src.zero();
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t k = baseB + (int64_t)u * 64 + lane;
        vB[u] = (wave > 0 && k < n) ? src.load(k) : src.zero();
    }
    double mfirstA[2], mfirstB[2];
    src.to_m((baseA > 0 && baseA - 1 < n) ? src.load(baseA - 1) : src.zero(), mfirstA);
    src.to_m((wave > 0 && baseB - 1 < n) ? src.load(baseB - 1) : src.zero(), mfirstB);
    // the same value in every lane: kept in scalar registers until the apply step needs it again
#pragma unroll
    for (int r = 0; r < 2; ++r) { mfirstA[r] = uw_lane63(mfirstA[r]); mfirstB[r] = uw_lane63(mfirstB[r]); }
    double mB[U][2];
    int ccB[U];
    // ---- chunk A: every wave
    {
        double mA[U][2];
        int ccA[U];
        int T[2];
        unsigned int nh;
        uw2_compute<SRC, U>(src, lane, baseA, n, vA, mfirstA, mA, ccA, T, &nh);
        if (lane == 0) { wsumA_s[0][wave] = T[0]; wsumA_s[1][wave] = T[1]; wnanA_s[wave] = (int)nh; }
#pragma unroll
        for (int u = 0; u < U; ++u) { mA_s[u][threadIdx.x] = make_double2(mA[u][0], mA[u][1]); ccA_s[u][threadIdx.x] = ccA[u]; }
    }
    __syncthreads();
    if (wave == 0) {
        // A's aggregate, A's look-back, A's inclusive prefix -- beside chunk B's arithmetic in the other waves
        const UwGather ga = uw2_gather(wsumA_s, wnanA_s, lane, 0, 0);
        const int T[2] = {ga.tot[0], ga.tot[1]};
        const unsigned int nan_here = ga.nan_all;
        if (lane == 0) uw_publish(&links[2 * t].agg, (unsigned)(T[0] + 8192) | ((unsigned)(T[1] + 8192) << 14) | (nan_here << 28), 1u);
        int E[2];
        unsigned int nan_before;
        bool gave_up;
        uw2_lookback<NROW>(links, 2 * t, lane, E, &nan_before, &gave_up);
        if (lane == 0) {
            const unsigned int nn = nan_before | nan_here;
            uw_publish(&links[2 * t].pre0, (unsigned)(E[0] + T[0]), 1u | ((nn & 1u) << 1));
            if (NROW == 2) uw_publish(&links[2 * t].pre1, (unsigned)(E[1] + T[1]), 1u | (((nn >> 1) & 1u) << 1));
            excl_s[0] = E[0]; excl_s[1] = E[1]; nanb_s = nan_before; gaveup_s = gave_up ? 1u : 0u;
        }
    } else {
        // ---- chunk B: waves 1..15; the last one to finish publishes B's aggregate
        int T[2];
        unsigned int nh;
        uw2_compute<SRC, U>(src, lane, baseB, n, vB, mfirstB, mB, ccB, T, &nh);
        unsigned int last = 0;
        if (lane == 0) {
            wsumB_s[0][wave] = T[0]; wsumB_s[1][wave] = T[1]; wnanB_s[wave] = (int)nh;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            last = atomicAdd(&doneB_s, 1u) == (unsigned)(NW - 2) ? 1u : 0u;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        if (__builtin_amdgcn_readfirstlane((int)last)) {
            const UwGather gb = uw2_gather(wsumB_s, wnanB_s, lane, 1, 0);
            if (lane == 0) uw_publish(&links[2 * t + 1].agg, (unsigned)(gb.tot[0] + 8192) | ((unsigned)(gb.tot[1] + 8192) << 14) | (gb.nan_all << 28), 1u);
        }
    }
    __syncthreads();
    // totals of A and B, and of the waves before this one in each
    const int EA[2] = {excl_s[0], excl_s[1]};
    const unsigned int nan_before = nanb_s;
    const UwGather ga = uw2_gather(wsumA_s, wnanA_s, lane, 0, wave);
    const UwGather gb = uw2_gather(wsumB_s, wnanB_s, lane, 1, wave);
    const int TA[2] = {ga.tot[0], ga.tot[1]}, inA[2] = {ga.before[0], ga.before[1]}, inB[2] = {gb.before[0], gb.before[1]};
    const unsigned int nanA = ga.nan_all, nanInA = ga.nan_before, nanInB = gb.nan_before;
    if (wave == 0 && lane == 0) {
        // B's inclusive prefix needs no look-back: A's carry + A's total + B's total
        const unsigned int nn = nan_before | nanA | gb.nan_all;
        uw_publish(&links[2 * t + 1].pre0, (unsigned)(EA[0] + TA[0] + gb.tot[0]), 1u | ((nn & 1u) << 1));
        if (NROW == 2) uw_publish(&links[2 * t + 1].pre1, (unsigned)(EA[1] + TA[1] + gb.tot[1]), 1u | (((nn >> 1) & 1u) << 1));
    }
    bool bad = gaveup_s != 0u;
    if (wave > 0)       // B first: its values are the ones in registers
        bad = uw2_apply<SRC, U>(src, out, lane, baseB, n, mB, ccB, mfirstB, EA[0] + TA[0] + inB[0], EA[1] + TA[1] + inB[1], nan_before | nanA | nanInB) || bad;
    {
        double mA[U][2];
        int ccA[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const double2 mm = mA_s[u][threadIdx.x]; mA[u][0] = mm.x; mA[u][1] = mm.y; ccA[u] = ccA_s[u][threadIdx.x]; }
        bad = uw2_apply<SRC, U>(src, out, lane, baseA, n, mA, ccA, mfirstA, EA[0] + inA[0], EA[1] + inA[1], nan_before | nanInA) || bad;
    }
    if (__any(bad) && lane == 0) atomicOr(flag, 1);
}

