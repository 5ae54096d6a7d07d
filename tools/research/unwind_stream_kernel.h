// tools/research/unwind_stream_kernel.h -- NOT part of the library.  Persistent, role-split forms of the one-pass unwind!
// (k_unwind_onepass in pixell.jl_amd/csrc/pxl_unwrap.h), built twice in round 4, bit-identical both times (33 GPU parity tests,
// > 1 000 fuzz cases each) and slower than what ships both times.
//
// Second form (this file; late round 4): wave 0 carries tickets, links and the look-back, waves 1..15 compute; a chunk's values wait
// in LDS (two buffers), sums / carries triple buffered, single-word links, lane-parallel gathers, the shipping kernel's arithmetic.
//   pix2sky!(safe=true), 1e8 points out of place, one MI355X (tools/research/r04_38.sh ... r04_42.sh):
//     k_unwind_onepass, 7 168-point chunks, two workgroups per CU (ships)                   0.70-0.71 ms
//     k_unwind_stream, 3 840-point chunks (U = 4), one persistent workgroup per CU:
//          look-back of 1 / 2 / 4 windows per round trip                                    0.78-0.79 / 0.96 / 1.03 ms   (single-word links)
//          1 920- and 960-point chunks (one or two workgroups per CU)                       2.6-3.0 / 6.3-7.6 ms
//     at 3e6 and 2e7 points the streaming form was 15-20 % FASTER than the then-shipping kernel (0.057 vs 0.070, 0.246 vs 0.287 ms)
// Why it loses on long batches: the persistent workgroups run in step, so the newest 256 chunks publish their aggregates together
// and every look-back reaches back a whole generation; its latency L obeys L = t / (1 - t / round) (t = one poll round trip), which
// exceeds the round itself once chunks are small -- and large chunks do not fit two LDS buffers.  With 15 compute waves per CU all
// in the same phase, memory latency is also hidden less well than by two independent 16-wave workgroups.
// First form (removed; registers instead of LDS, three-word links): 0.98-1.39 ms over six chunk / workgroup shapes.
// To use: paste after k_unwind_onepass; launch grid = min(chunks, CUs), block = 64 * PXL_UWS_NW; links sized for PXL_UWS_CHUNK.
//
// ------------------------------------------------------------------------------------------------
// STREAMING one-pass form: the same links and the same arithmetic as k_unwind_onepass, with the look-back taken out of the
// workgroup's critical path.  One PERSISTENT workgroup per CU, split into roles: wave 0 is the carrier -- it takes the tickets,
// publishes aggregates and prefixes and looks back --, waves 1..15 compute.  One workgroup barrier per chunk:
//      compute waves:  C(k) -> park in LDS | barrier | A(k-1) from LDS     C(k+1) | barrier | A(k) ...
//      carrier wave:                       | barrier | aggregate(k), look-back(k), prefix(k), carry(k) into LDS | barrier | ...
// C = load + exact rewind + increments + sums, A = verify + store.  The look-back for chunk k runs beside A(k-1) and C(k+1): a
// compute wave waits for it only when it is slower than a whole round of its own work.  A chunk's rewound values wait in LDS
// (17 B per point, two buffers), so the registers hold one chunk at a time.  The persistent workgroups run in step, so the 256
// newest chunks publish their aggregates together and a look-back reaches back up to 256 links: the carrier reads FOUR 64-link
// windows per round trip (it has nothing else to do, and the registers of a one-workgroup-per-CU kernel to do it with).
// Sums, carries and tickets are triple / double buffered in LDS: a slot is rewritten only after a barrier that every reader of its
// previous content has passed.  Deadlock freedom as before: tickets; an aggregate is published before anything is waited for;
// every wait is for links with smaller ids; polls are bounded.
// ------------------------------------------------------------------------------------------------
#ifndef PXL_UWS_U
#define PXL_UWS_U 4
#endif
#define PXL_UWS_NW 16
#define PXL_UWS_NC (PXL_UWS_NW - 1)
#define PXL_UWS_CHUNK (64LL * PXL_UWS_U * PXL_UWS_NC)
#ifndef PXL_UWS_WIN
#define PXL_UWS_WIN 4
#endif
template <class SRC>
__global__ __launch_bounds__(64 * PXL_UWS_NW) void k_unwind_stream(SRC src, typename SRC::raw_t* out, int64_t n, int64_t nchunks,
                                                                  UwLink* __restrict__ links, unsigned int* __restrict__ ticket,
                                                                  int32_t* __restrict__ flag) {
    constexpr int U = PXL_UWS_U, NROW = SRC::NROW, NW = PXL_UWS_NW, NC = PXL_UWS_NC, WIN = PXL_UWS_WIN;
    __shared__ unsigned int id_s[2];
    __shared__ int wsum_s[3][2][NW], wnan_s[3][NW];
    __shared__ int E_s[3][2];
    __shared__ unsigned int nanb_s[3], gaveup_s[3];
    __shared__ double2 mL_s[2][U][64 * NC];
    __shared__ unsigned char ccL_s[2][U][64 * NC];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slot = (wave - 1) * 64 + lane;                       // compute waves: this lane's place in a parked group
    const double P = src.period, rP = src.rperiod, ref = src.ref;
    if (threadIdx.x == 0) id_s[0] = atomicAdd(ticket, 1u);
    if (threadIdx.x < 3) { wsum_s[threadIdx.x][0][0] = 0; wsum_s[threadIdx.x][1][0] = 0; wnan_s[threadIdx.x][0] = 0; }     // wave 0 holds no points
    __syncthreads();
    // the chunk waiting for its carry (compute waves): where it starts, the m before its first point, its own NaN flag
    bool has_p = false;
    int64_t base_p = 0;
    double mfirst_p[2] = {0.0, 0.0};
    unsigned int nh_p = 0;
    bool bad = false;
    for (int it = 0;; ++it) {
        const int par = it & 1, cur = it % 3, prev = (it + 2) % 3;
        const int64_t id = (int64_t)(unsigned int)__builtin_amdgcn_readfirstlane((int)id_s[par]);
        const bool has_new = id < nchunks;
        if (!has_new && !has_p) break;                             // the same in every wave
        if (threadIdx.x == 0) id_s[par ^ 1] = atomicAdd(ticket, 1u);            // next round's ticket (its slot was last read before the previous barrier)
        const int64_t base_n = id * PXL_UWS_CHUNK + (int64_t)(wave - 1) * 64 * U;
        double mfirst_n[2] = {0.0, 0.0};
        unsigned int nh_n = 0;
        if (has_new && wave > 0) {
            // ---- C: load, exact rewind, increments, this wave's sums; the values go to LDS
            const int64_t left = n - base_n;
            const int nvalid = left <= 0 ? 0 : (left >= 64 * U ? 64 * U : (int)left);
            const bool kpos = base_n > 0;
            typename SRC::raw_t v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = u * 64 + lane;
                v[u] = (idx < nvalid) ? src.load(base_n + idx) : src.zero();
            }
            src.to_m((base_n > 0 && base_n - 1 < n) ? src.load(base_n - 1) : src.zero(), mfirst_n);
            mfirst_n[0] = uw_lane63(mfirst_n[0]); mfirst_n[1] = uw_lane63(mfirst_n[1]);          // the same in every lane: scalar
            double mlast[2] = {mfirst_n[0], mfirst_n[1]};
            int sum[2] = {0, 0};
            bool nanl[2] = {false, false};
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = u * 64 + lane;
                const bool valid = idx < nvalid;
                double mm[2], mp[2];
                int c[2] = {0, 0};
                uw_element(src, lane, (int64_t)((kpos || idx > 0) ? 1 : 0), valid, v[u], mlast, mm, mp, c);
                mL_s[par][u][slot] = make_double2(mm[0], mm[1]);
                ccL_s[par][u][slot] = (unsigned char)((c[0] + 1) | (NROW == 2 ? (c[1] + 1) << 2 : 0));
#pragma unroll
                for (int r = 0; r < NROW; ++r) {
                    sum[r] += c[r];
                    nanl[r] = nanl[r] || (valid && mm[r] != mm[r]);
                    mlast[r] = uw_lane63(mm[r]);
                }
            }
            nh_n = (__ballot(nanl[0]) != 0ull ? 1u : 0u) | (NROW == 2 && __ballot(nanl[1]) != 0ull ? 2u : 0u);
            const int T0 = uw_wave_total(sum[0]), T1 = NROW == 2 ? uw_wave_total(sum[1]) : 0;
            if (lane == 0) { wsum_s[cur][0][wave] = T0; wsum_s[cur][1][wave] = T1; wnan_s[cur][wave] = (int)nh_n; }
        }
        __syncthreads();       // chunk `it`: sums complete.  Chunk it - 1: its carry is in E_s[prev] (the carrier wrote it before arriving here)
        if (wave == 0) {
            if (has_new) {
                const UwGather gt = uw_gather<NW>(wsum_s[cur], wnan_s[cur], lane, 0);
                const int T[2] = {gt.tot[0], gt.tot[1]};
                const unsigned int nan_here = gt.nan_all;
                if (lane == 0) uw_publish(&links[id], T[0], T[1], nan_here, 1u);
                int E[2];
                unsigned int nan_before;
                bool gave_up;
                uw_lookback<WIN>(links, id, lane, E, &nan_before, &gave_up);
                if (lane == 0) {
                    uw_publish(&links[id], E[0] + T[0], E[1] + T[1], nan_before | nan_here, 2u);
                    E_s[cur][0] = E[0]; E_s[cur][1] = E[1]; nanb_s[cur] = nan_before; gaveup_s[cur] = gave_up ? 1u : 0u;
                }
            }
        } else {
            if (has_p) {
                // ---- A: the chunk of the previous round, from LDS buffer par ^ 1; k_unwind_onepass's apply step
                const UwGather gi = uw_gather<NW>(wsum_s[prev], wnan_s[prev], lane, wave);
                int carry[2] = {E_s[prev][0] + gi.before[0], E_s[prev][1] + gi.before[1]};
                const unsigned int nan_before = nanb_s[prev] | gi.nan_before;
                bool pex[2] = {(nan_before & 1u) != 0, (nan_before & 2u) != 0};
                bad = bad || gaveup_s[prev] != 0u;
                double ylast[2] = {mfirst_p[0] - (double)carry[0] * P, mfirst_p[1] - (double)carry[1] * P};
                const bool full = base_p > 0 && base_p + 64 * U <= n;
                auto apply_group = [&](auto plain_tag, int u) {
                    constexpr bool PLAIN = decltype(plain_tag)::value;     // no NaN here or before, every point exists, none is the first
                    const int64_t k = base_p + (int64_t)u * 64 + lane;
                    const bool valid = PLAIN || k < n;
                    const double2 mm = mL_s[par ^ 1][u][slot];
                    const int cb = (int)ccL_s[par ^ 1][u][slot];
                    const double mv[2] = {mm.x, mm.y};
                    const int ccv = (cb & 3) | ((cb >> 2) << 16);
                    const int s = uw_scan64(ccv);
                    const int tot = __builtin_amdgcn_readlane(s, 63);
                    double y[2] = {0.0, 0.0};
#pragma unroll
                    for (int r = 0; r < NROW; ++r) {
                        const int field = r == 0 ? (s & 0xffff) : (s >> 16);
                        const int rr = carry[r] + field - (lane + 1);
                        carry[r] += (r == 0 ? (tot & 0xffff) : (tot >> 16)) - 64;
                        const double rrd = (double)rr;
                        const double yq = mv[r] - rrd * P;
                        const double yprev = uw_shr1_first(yq, ylast[r]);
                        ylast[r] = uw_lane63(yq);
                        bool poisoned = false;
                        if (!PLAIN) {
                            const unsigned long long nanmask = __ballot(valid && mv[r] != mv[r]);
                            poisoned = pex[r] || (nanmask & ((2ull << lane) - 1ull)) != 0ull;
                            pex[r] = pex[r] || nanmask != 0ull;
                        }
                        if (!valid) continue;
                        if (poisoned) { y[r] = __builtin_nan("") + ref; continue; }
                        if (PLAIN || k > 0) {
                            const double a = mv[r] - yprev;
                            const double qa = a * rP;
                            if (!(fabs(qa - rrd) < 0.4999)) {
                                const double q = a / P;
                                if (!(rint(q) == rrd)) bad = true;
                            }
                        }
                        y[r] = yq + ref;
                    }
                    if (valid) SRC::store(out, k, y);
                };
                if ((nan_before | nh_p) == 0u && full) {
#pragma unroll
                    for (int u = 0; u < U; ++u) apply_group(std::true_type{}, u);
                } else {
#pragma unroll
                    for (int u = 0; u < U; ++u) apply_group(std::false_type{}, u);
                }
            }
            base_p = base_n; mfirst_p[0] = mfirst_n[0]; mfirst_p[1] = mfirst_n[1]; nh_p = nh_n;
        }
        has_p = has_new;
    }
    if (__any(bad) && lane == 0) atomicOr(flag, 1);
}

