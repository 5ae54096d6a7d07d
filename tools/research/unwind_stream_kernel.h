// tools/research/unwind_stream_kernel.h -- NOT part of the library.  A persistent, role-split form of the one-pass unwind!
// (k_unwind_onepass in pixell.jl_amd/csrc/pxl_unwrap.h), built and measured in round 4 and dropped: it gives the same bits
// (32 GPU parity tests, 776 fuzz cases) and is SLOWER.
//
//   pix2sky!(safe=true), 1e8 points out of place, MI355X (tools/prof_unwind.py; tools/research/r04_17.sh, r04_18.sh):
//     k_unwind_onepass (16 waves x 4 x 64 points per workgroup, one link per workgroup)      0.897-0.905 ms
//     k_unwind_stream  NW=16 U=4 (90 VGPRs, one workgroup per CU)                              0.984-0.999 ms
//                      NW=16 U=2 (65 VGPRs)                                                    1.21 ms
//                      NW=16 U=8 (128 VGPRs)                                                   1.39 ms
//                      NW=8  U=4                                                               1.30-1.38 ms
//                      NW=8  U=8                                                               1.08-1.09 ms
//                      NW=4  U=8                                                               1.45-2.0 ms
//   (workgroups per CU 1 / 2 / 4 requested: no difference where the registers allow only one)
//
// Why it loses: the carrier wave does take the look-back out of the compute waves' way, but a compute wave now holds TWO chunks
// of rewound values (90 VGPRs -> one 16-wave workgroup per CU instead of two), every round still has a workgroup barrier, and the
// rounds of the 256 persistent workgroups run in lock step, so each generation's 256 aggregates appear together and every
// look-back walks ~4 windows of aggregates before it meets a prefix.  To use: paste into pxl_unwrap.h after k_unwind_onepass and
// launch with grid = min(chunks, CUs), block = 64 * PXL_UWS_NW, links sized for chunks of 64 * PXL_UWS_U * (PXL_UWS_NW - 1) points.
//
// ------------------------------------------------------------------------------------------------
// STREAMING one-pass form (round 4, late): the same links and the same arithmetic, with the look-back taken OUT of the workgroup's
// critical path.  In k_unwind_onepass the 16 waves of a workgroup sit at a barrier while wave 0 polls (a CU holds two such
// workgroups: 0.88 ms per 1e8 points for 0.36 ms of arithmetic and 0.52 ms of memory time).  Here a workgroup is PERSISTENT and
// split into roles: wave 0 is the carrier -- it takes the tickets, publishes aggregates and prefixes and does the look-back --
// and waves 1..15 compute.  With ONE workgroup barrier per chunk:
//      compute waves:  C(k+1) | barrier | A(k)      C(k+2) | barrier | A(k+1) ...       C = load + exact rewind + sums, A = verify + store
//      carrier wave:          | barrier | publish agg(k+1), look back for k+1, leave E(k+1) in LDS | barrier | ...
// so the look-back for chunk k+1 runs beside the apply step of chunk k and the compute step of chunk k+2, and a compute wave only
// ever waits at the barrier when the look-back is slower than its own work.  A compute wave keeps the rewound values of two
// chunks in registers (the one waiting for its carry and the one just computed).  Chunk = 15 waves x 64 U points.  Deadlock
// freedom as before: a ticket holder publishes its aggregate right after the barrier that completes it, before it waits for
// anything; every wait is for links with smaller tickets.
// ------------------------------------------------------------------------------------------------
#ifndef PXL_UWS_U
#define PXL_UWS_U 4
#endif
#ifndef PXL_UWS_NW
#define PXL_UWS_NW 16
#endif
#define PXL_UWS_NC (PXL_UWS_NW - 1)
template <class SRC>
__global__ __launch_bounds__(64 * PXL_UWS_NW) void k_unwind_stream(SRC src, typename SRC::raw_t* out, int64_t n, int64_t nchunks,
                                                                  UwLink* __restrict__ links, unsigned int* __restrict__ ticket,
                                                                  int32_t* __restrict__ flag) {
    constexpr int U = PXL_UWS_U, NROW = SRC::NROW, NW = PXL_UWS_NW, NC = PXL_UWS_NC;
    constexpr int64_t CH = (int64_t)NC * 64 * U;                 // points per chunk
    __shared__ unsigned int id_s[2];
    __shared__ int wsum_s[2][2][NW], wnan_s[2][NW];              // [parity][row][wave]
    __shared__ int E_s[2][2];
    __shared__ unsigned int nanb_s[2], gaveup_s[2];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const double P = src.period, rP = src.rperiod, ref = src.ref;
    if (threadIdx.x == 0) id_s[0] = atomicAdd(ticket, 1u);
    __syncthreads();
    // the chunk waiting for its carry (compute waves only)
    double m_p[U][2], mfirst_p[2] = {0.0, 0.0};
    int cc_p[U], cin_p[2] = {0, 0};
    unsigned int nanin_p = 0;
    int64_t base_p = 0;
    bool has_p = false;
    bool bad = false;
#pragma unroll
    for (int u = 0; u < U; ++u) { m_p[u][0] = m_p[u][1] = 0.0; cc_p[u] = 0; }
    for (int it = 0;; ++it) {
        const int par = it & 1;
        const int64_t id = (int64_t)id_s[par];
        const bool has_new = id < nchunks;
        if (!has_new && !has_p) break;                           // uniform: id and has_p are the same in every wave
        if (threadIdx.x == 0) id_s[par ^ 1] = atomicAdd(ticket, 1u);          // the ticket of the next round, seen after the barrier below
        double m_n[U][2], mfirst_n[2] = {0.0, 0.0};
        int cc_n[U];
        const int64_t base_n = id * CH + (int64_t)(wave - 1) * 64 * U;
#pragma unroll
        for (int u = 0; u < U; ++u) { m_n[u][0] = m_n[u][1] = 0.0; cc_n[u] = 0; }
        if (has_new && wave > 0) {
            // ---- C: load, exact rewind, nominal increments, this wave's sums
            typename SRC::raw_t v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t k = base_n + (int64_t)u * 64 + lane;
                v[u] = (k < n) ? src.load(k) : src.zero();
            }
            src.to_m((base_n > 0 && base_n - 1 < n) ? src.load(base_n - 1) : src.zero(), mfirst_n);
            double mlast[2] = {mfirst_n[0], mfirst_n[1]};
            int sum[2] = {0, 0};
            bool nanl[2] = {false, false};
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t k = base_n + (int64_t)u * 64 + lane;
                double mp[2];
                int c[2] = {0, 0};
                uw_element(src, lane, k, k < n, v[u], mlast, m_n[u], mp, c);
                cc_n[u] = (c[0] + 1) | (NROW == 2 ? (c[1] + 1) << 16 : 0);
#pragma unroll
                for (int r = 0; r < NROW; ++r) {
                    sum[r] += c[r];
                    nanl[r] = nanl[r] || (k < n && m_n[u][r] != m_n[u][r]);
                    mlast[r] = uw_lane63(m_n[u][r]);
                }
            }
            const int T0 = uw_wave_total(sum[0]), T1 = NROW == 2 ? uw_wave_total(sum[1]) : 0;
            const unsigned int nh = (__ballot(nanl[0]) != 0ull ? 1u : 0u) | (NROW == 2 && __ballot(nanl[1]) != 0ull ? 2u : 0u);
            if (lane == 0) { wsum_s[par][0][wave] = T0; wsum_s[par][1][wave] = T1; wnan_s[par][wave] = (int)nh; }
        }
        __syncthreads();     // the sums of chunk `it` are complete; the carrier has left E of chunk it - 1 in E_s[par ^ 1]
        if (wave == 0) {
            if (has_new) {
                int T[2] = {0, 0};
                unsigned int nan_here = 0;
#pragma unroll
                for (int w2 = 1; w2 < NW; ++w2) { T[0] += wsum_s[par][0][w2]; T[1] += wsum_s[par][1][w2]; nan_here |= (unsigned)wnan_s[par][w2]; }
                if (lane == 0) uw_publish(&links[id].agg, (unsigned)(T[0] + 8192) | ((unsigned)(T[1] + 8192) << 14) | (nan_here << 28), 1u);
                int E[2] = {0, 0};
                unsigned int nan_before = 0;
                bool gave_up = false;
                if (id > 0) {
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
                        const bool hasP = j < 0 || (((unsigned)p0 & 1u) && ((unsigned)p1 & 1u));
                        const bool hasA = j < 0 || ((unsigned)a & 1u);
                        const unsigned long long pmask = __ballot(hasP), amask = __ballot(hasA);
                        const int first = pmask ? __builtin_ctzll(pmask) : 64;
                        const unsigned long long need = first >= 64 ? ~0ull : ((1ull << first) - 1ull);
                        if ((amask & need) != need) {
                            if (++polls > (1u << 22)) { gave_up = true; break; }
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
                        if (__ballot(nb & 1u) != 0ull) nan_before |= 1u;
                        if (__ballot(nb & 2u) != 0ull) nan_before |= 2u;
                        if (first < 64) break;
                        top -= 64;
                        polls = 0;
                    }
                }
                if (lane == 0) {
                    const unsigned int nn = nan_before | nan_here;
                    uw_publish(&links[id].pre0, (unsigned)(E[0] + T[0]), 1u | ((nn & 1u) << 1));
                    if (NROW == 2) uw_publish(&links[id].pre1, (unsigned)(E[1] + T[1]), 1u | (((nn >> 1) & 1u) << 1));
                    E_s[par][0] = E[0]; E_s[par][1] = E[1]; nanb_s[par] = nan_before; gaveup_s[par] = gave_up ? 1u : 0u;
                }
            }
        } else {
            // this wave's carry INSIDE the new chunk (sums of the compute waves before it): taken now, because the buffer is
            // rewritten two rounds from here while slower waves may still be applying
            int cin_n[2] = {0, 0};
            unsigned int nanin_n = 0;
            if (has_new)
                for (int w2 = 1; w2 < wave; ++w2) { cin_n[0] += wsum_s[par][0][w2]; cin_n[1] += wsum_s[par][1][w2]; nanin_n |= (unsigned)wnan_s[par][w2]; }
            if (has_p) {
                // ---- A: the chunk of the previous round: its carry is in E_s[par ^ 1]
                int carry[2] = {E_s[par ^ 1][0] + cin_p[0], E_s[par ^ 1][1] + cin_p[1]};
                const unsigned int nan_before = nanb_s[par ^ 1] | nanin_p;
                bool pex[2] = {(nan_before & 1u) != 0, (nan_before & 2u) != 0};
                bad = bad || gaveup_s[par ^ 1] != 0u;
                double mlast[2] = {mfirst_p[0], mfirst_p[1]};
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int64_t k = base_p + (int64_t)u * 64 + lane;
                    const bool valid = k < n;
                    const int sc = uw_scan64(cc_p[u]);
                    const int tot = __builtin_amdgcn_readlane(sc, 63);
                    double y[2] = {0.0, 0.0};
#pragma unroll
                    for (int r = 0; r < NROW; ++r) {
                        const double mp = uw_shr1_first(m_p[u][r], mlast[r]);
                        mlast[r] = uw_lane63(m_p[u][r]);
                        const int field = r == 0 ? (sc & 0xffff) : (sc >> 16);
                        const int c = (r == 0 ? (cc_p[u] & 0xffff) : (cc_p[u] >> 16)) - 1;
                        const int rr = carry[r] + field - (lane + 1);            // r_k
                        carry[r] += (r == 0 ? (tot & 0xffff) : (tot >> 16)) - 64;
                        const unsigned long long nanmask = __ballot(valid && m_p[u][r] != m_p[u][r]);
                        const bool poisoned = pex[r] || (nanmask & ((2ull << lane) - 1ull)) != 0ull;
                        pex[r] = pex[r] || nanmask != 0ull;
                        if (!valid) continue;
                        if (poisoned) { y[r] = __builtin_nan("") + ref; continue; }
                        if (k > 0) {
                            const double yprev = mp - (double)(rr - c) * P;       // y[k-1] as the reference forms it
                            const double a = m_p[u][r] - yprev;
                            const double qa = a * rP;
                            if (!(fabs(qa - (double)rr) < 0.4999)) {
                                const double q = a / P;
                                if (!(rint(q) == (double)rr)) bad = true;
                            }
                        }
                        y[r] = (m_p[u][r] - (double)(k > 0 ? rr : 0) * P) + ref;     // k = 0: m - 0 = m, bit for bit
                    }
                    if (valid) SRC::store(out, k, y);
                }
            }
            // the new chunk becomes the pending one
#pragma unroll
            for (int u = 0; u < U; ++u) { m_p[u][0] = m_n[u][0]; m_p[u][1] = m_n[u][1]; cc_p[u] = cc_n[u]; }
            mfirst_p[0] = mfirst_n[0]; mfirst_p[1] = mfirst_n[1];
            cin_p[0] = cin_n[0]; cin_p[1] = cin_n[1]; nanin_p = nanin_n; base_p = base_n;
        }
        has_p = has_new;
    }
    if (__any(bad) && lane == 0) atomicOr(flag, 1);
}

