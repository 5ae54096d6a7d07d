// pxl_place.h -- class-aware placement of a (source, destination) pair inside one allocation: the native twin of
// pixell.jl_amd/placement.py::place_pair, so that a C or Julia host gets the placement through the ABI (pxl_mem_pair_alloc /
// pxl_mem_pair_free in include/pixell_hip.h) instead of re-implementing it over pxl_mem_probe_pair.  Host code only; included by
// pxl_kernels.hip after the probe entry.  What the classes are and why a destination across a boundary matters: pxl_spread.h.
#pragma once
#include <algorithm>
#include <vector>

namespace pxl_place {

static const uint64_t GiB = 1ull << 30;
static const uint64_t AL = 2ull << 20;                 // offsets are multiples of 2 MiB

struct ClassMap {
    std::vector<uint64_t> offs;                        // 1 GiB windows every `step` bytes
    std::vector<int> labels;
    int classes = 1, probes = 0;
};

// label the windows: two windows share a label when the eight-front store probe between them runs at the slow one of its rates
static int map_classes(char* base, uint64_t nbytes, uint64_t step, hipStream_t st, ClassMap* cm) {
    const uint64_t window = GiB;
    cm->offs.clear(); cm->labels.clear(); cm->classes = 1; cm->probes = 0;
    for (uint64_t o = 0; o + window <= nbytes; o += step) cm->offs.push_back(o);
    const size_t nw = cm->offs.size();
    cm->labels.assign(nw, 0);
    if (nw < 2) return PXL_OK;
    std::vector<float> t0(nw, 0.f);
    float lo = 1e30f, hi = 0.f;
    for (size_t k = 1; k < nw; ++k) {                  // calibration: every window against window 0 (bimodal times)
        int rc = pxl_mem_probe_pair(base + cm->offs[k], base + cm->offs[0], window, 3, &t0[k], st);
        if (rc) return rc;
        ++cm->probes;
        lo = std::min(lo, t0[k]); hi = std::max(hi, t0[k]);
    }
    // one rate only: the part's two absolute rates, 5.6-5.9 TB/s within a class and 6.6-7.0 TB/s across, divide at 6.25 TB/s
    const float thr = hi >= 1.10f * lo ? sqrtf(lo * hi) : (float)(2.0 * window / 6.25e6);
    std::vector<size_t> refs{0};
    for (size_t k = 1; k < nw; ++k) cm->labels[k] = t0[k] > thr ? 0 : -1;
    for (size_t k = 1; k < nw; ++k) {
        if (cm->labels[k] >= 0) continue;
        for (size_t c = 1; c < refs.size() && cm->labels[k] < 0; ++c) {
            float t;
            int rc = pxl_mem_probe_pair(base + cm->offs[k], base + cm->offs[refs[c]], window, 3, &t, st);
            if (rc) return rc;
            ++cm->probes;
            if (t > thr) cm->labels[k] = (int)c;
        }
        if (cm->labels[k] < 0) { cm->labels[k] = (int)refs.size(); refs.push_back(k); }
    }
    cm->classes = (int)refs.size();
    return PXL_OK;
}

// does [o, o + n) overlap a window of a class in `used` (bit mask)?  returns the share of its windows that do NOT
static double foreign_share(const ClassMap& cm, uint64_t o, uint64_t n, unsigned used) {
    int tot = 0, fr = 0;
    for (size_t k = 0; k < cm.offs.size(); ++k)
        if (cm.offs[k] + GiB > o && cm.offs[k] < o + n) { ++tot; if (!((used >> (cm.labels[k] & 31)) & 1u)) ++fr; }
    return tot ? (double)fr / tot : 1.0;
}

}  // namespace pxl_place

int pxl_mem_pair_alloc(uint64_t src_bytes, uint64_t dst_bytes, uint64_t headroom_bytes, pxl_mem_pair* out, void* stream) {
    using namespace pxl_place;
    if (!out) return fail(PXL_EINVAL, "mem_pair_alloc: null result");
    memset(out, 0, sizeof *out);
    if (src_bytes == 0 || dst_bytes == 0) return fail(PXL_EINVAL, "mem_pair_alloc: empty map");
    hipStream_t st = (hipStream_t)stream;
    const uint64_t bs = (src_bytes + AL - 1) / AL * AL, bd = (dst_bytes + AL - 1) / AL * AL;
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    uint64_t total = bs + bd + headroom_bytes;
    const uint64_t cap = free_b > 6 * GiB ? free_b - 6 * GiB : 0;
    if (total > cap) total = std::max<uint64_t>(bs + bd + AL, cap);
    total = total / AL * AL;
    char* arena = nullptr;
    hipError_t e = hipMalloc((void**)&arena, total);
    if (e != hipSuccess) return fail(PXL_EHIP, "mem_pair_alloc: hipMalloc of %.1f GiB: %s", total / (double)GiB, hipGetErrorString(e));
    ClassMap cm;
    const uint64_t step = 2 * GiB;
    int rc = total >= 2 * GiB ? map_classes(arena, total, step, st, &cm) : PXL_OK;
    if (rc) { (void)hipFree(arena); return rc; }
    // boundaries between runs of equal label (half-way between the last window of one run and the first of the next); the best
    // one has the most room of its two classes on either side, up to half the destination
    uint64_t best_b = 0, best_score = 0;
    {
        std::vector<uint64_t> bounds;
        for (size_t k = 1; k < cm.offs.size(); ++k)
            if (cm.labels[k] != cm.labels[k - 1]) bounds.push_back((cm.offs[k - 1] + cm.offs[k] + GiB) / 2);
        for (size_t i = 0; i < bounds.size(); ++i) {
            const uint64_t left = bounds[i] - (i ? bounds[i - 1] : 0), right = (i + 1 < bounds.size() ? bounds[i + 1] : total) - bounds[i];
            const uint64_t score = std::min(std::min(left, right), bd / 2);
            if (score > best_score) { best_score = score; best_b = bounds[i]; }
        }
    }
    uint64_t dst_off = (total - bd) / AL * AL, src_off = 0;
    int two = 0, own = 0;
    if (best_score >= std::min<uint64_t>(bd / 8, GiB) && best_score > 0) {
        const uint64_t half = bd / 2;
        dst_off = best_b > half ? best_b - half : 0;
        dst_off = std::min(dst_off, total - bd) / AL * AL;
        two = 1;
        unsigned used = 0;
        for (size_t k = 0; k < cm.offs.size(); ++k)
            if (cm.offs[k] + GiB > dst_off && cm.offs[k] < dst_off + bd) used |= 1u << (cm.labels[k] & 31);
        // the source: the free stretch with the largest share of windows in a class the destination does not touch, far from it
        double best_share = -1.0; uint64_t best_dist = 0; bool have = false;
        auto consider = [&](uint64_t o) {
            o = o / AL * AL;
            if (!(o + bs <= dst_off || dst_off + bd <= o) || o + bs > total) return;
            const double sh = floor(foreign_share(cm, o, bs, used) * 100.0 + 0.5) / 100.0;
            const uint64_t dist = o > dst_off ? o - dst_off : dst_off - o;
            if (!have || sh > best_share || (sh == best_share && dist > best_dist)) { have = true; best_share = sh; best_dist = dist; src_off = o; }
        };
        if (dst_off >= bs) { for (uint64_t o = 0; o + bs <= dst_off; o += step) consider(o); consider(dst_off - bs); }
        if (total - (dst_off + bd) >= bs) { for (uint64_t o = (dst_off + bd + AL - 1) / AL * AL; o + bs <= total; o += step) consider(o); consider(total - bs); }
        if (!have) { (void)hipFree(arena); return fail(PXL_EINVAL, "mem_pair_alloc: head-room too small to hold the source beside the destination"); }
        own = best_share >= 0.995;
    }
    // no third class inside the allocation: look for the source in separate allocations (rejected candidates are held as ballast
    // while the search goes on and freed before returning), within 96 GiB and the free memory
    char* src_alloc = nullptr;
    int tried = 0;
    if (two && !own && bs >= GiB) {
        std::vector<char*> refs;
        {
            unsigned seen = 0;
            for (size_t k = 0; k < cm.offs.size(); ++k)
                if (cm.offs[k] + GiB > dst_off && cm.offs[k] < dst_off + bd && !((seen >> (cm.labels[k] & 31)) & 1u)) { seen |= 1u << (cm.labels[k] & 31); refs.push_back(arena + cm.offs[k]); }
        }
        const float thr = (float)(2.0 * GiB / 6.25e6);
        std::vector<char*> ballast;
        uint64_t held = 0;
        while (rc == PXL_OK) {
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < bs + 8 * GiB || held + bs > 96 * GiB) break;
            char* cand = nullptr;
            if (hipMalloc((void**)&cand, bs) != hipSuccess) { (void)hipGetLastError(); break; }
            ++tried;
            bool same = false;
            for (char* r : refs) {
                float t;
                rc = pxl_mem_probe_pair(cand, r, GiB, 3, &t, st);
                if (rc) break;
                ++cm.probes;
                same = same || t > thr;
            }
            if (rc == PXL_OK && !same) { src_alloc = cand; own = 1; break; }
            ballast.push_back(cand);
            held += bs;
        }
        for (char* b : ballast) (void)hipFree(b);
        if (rc) { (void)hipFree(arena); return rc; }
    }
    char* src = src_alloc ? src_alloc : arena + src_off;
    e = hipMemsetAsync(src, 0, src_bytes, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { if (src_alloc) (void)hipFree(src_alloc); (void)hipFree(arena); return fail(PXL_EHIP, "mem_pair_alloc: %s", hipGetErrorString(e)); }
    out->src = src; out->dst = arena + dst_off; out->arena = arena; out->src_alloc = src_alloc;
    out->arena_bytes = total; out->src_offset = src_alloc ? 0 : src_off; out->dst_offset = dst_off;
    out->classes = cm.classes; out->dst_two_classes = two; out->src_own_class = own; out->probes = cm.probes;
    out->separate_tried = tried;
    return PXL_OK;
}

int pxl_mem_pair_free(pxl_mem_pair* p) {
    if (!p) return fail(PXL_EINVAL, "mem_pair_free: null");
    hipError_t e1 = p->src_alloc ? hipFree(p->src_alloc) : hipSuccess, e2 = p->arena ? hipFree(p->arena) : hipSuccess;
    memset(p, 0, sizeof *p);
    if (e1 != hipSuccess || e2 != hipSuccess) return fail(PXL_EHIP, "mem_pair_free: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
    return PXL_OK;
}

// ---- the default policy for ONE map-sized buffer: two classes, no head-room (include/pixell_hip.h)
int pxl_mem_alloc_placed(uint64_t bytes, uint64_t budget_bytes, void** out, pxl_mem_placed_info* info, void* stream) {
    using namespace pxl_place;
    if (!out) return fail(PXL_EINVAL, "mem_alloc_placed: null result");
    *out = nullptr;
    if (info) memset(info, 0, sizeof *info);
    if (bytes == 0) return fail(PXL_EINVAL, "mem_alloc_placed: empty buffer");
    hipStream_t st = (hipStream_t)stream;
    if (budget_bytes == 0) budget_bytes = 96 * GiB;
    auto alloc = [&](char** p) -> int {
        hipError_t e = hipMalloc((void**)p, bytes);
        if (e != hipSuccess) { (void)hipGetLastError(); return fail(PXL_EHIP, "mem_alloc_placed: hipMalloc of %.1f GiB: %s", bytes / (double)GiB, hipGetErrorString(e)); }
        return PXL_OK;
    };
    char* best = nullptr;
    int best_share = -1, tries = 0, probes = 0;
    uint64_t held = 0, peak = 0;
    std::vector<char*> ballast;
    int rc = PXL_OK;
    if (bytes < 3 * GiB) {
        rc = alloc(&best);
        if (rc) return rc;
        best_share = 0; tries = 1;
    }
    while (bytes >= 3 * GiB) {
        char* cand = nullptr;
        rc = alloc(&cand);
        if (rc) { if (best) { rc = PXL_OK; } break; }             // out of memory mid-search: keep the best so far
        ++tries;
        ClassMap cm;
        rc = map_classes(cand, bytes, GiB, st, &cm);
        if (rc) { (void)hipFree(cand); break; }
        probes += cm.probes;
        int cnt[64] = {0}, major = 0;
        for (int l : cm.labels) { if (l >= 0 && l < 64 && ++cnt[l] > major) major = cnt[l]; }
        const int share = cm.labels.empty() ? 0 : (int)(100 - 100 * (int64_t)major / (int64_t)cm.labels.size());
        if (share > best_share) { if (best) { ballast.push_back(best); held += bytes; } best = cand; best_share = share; }
        else { ballast.push_back(cand); held += bytes; }
        peak = std::max(peak, held);
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = 0; }
        // (16 GiB and more: the driver builds such a buffer from several blocks and every candidate shows ~20 % in a second class)
        if (best_share >= 30 || (bytes >= 16 * GiB && best_share >= 20) || tries >= 24 || held + bytes > budget_bytes || free_b < bytes + 8 * GiB) break;
    }
    for (char* b : ballast) (void)hipFree(b);
    if (rc) { if (best) (void)hipFree(best); return rc; }
    *out = best;
    if (info) { info->tries = tries; info->probes = probes; info->two_classes = best_share >= 20; info->minor_share_pct = best_share < 0 ? 0 : best_share; info->ballast_bytes = peak; }
    return PXL_OK;
}

int pxl_mem_free(void* ptr) {
    if (!ptr) return PXL_OK;
    hipError_t e = hipFree(ptr);
    if (e != hipSuccess) return fail(PXL_EHIP, "mem_free: %s", hipGetErrorString(e));
    return PXL_OK;
}
