// pxl_kernels.hip -- gfx950 kernels + C ABI of libpixell_hip.so (see include/pixell_hip.h).
//
// Every kernel here is HBM-bound (streams or gathers of Float64); none is GEMM-shaped, so there is
// no MFMA.  The design rules are the memory ones: 16 B per lane coalesced loads/stores, source rows
// staged through LDS once per output tile, XCD-aware tile order so neighbouring tiles share an L2.
//
// One translation unit; the kernels live in topical headers included below:
//   pxl_device.h         shared device arithmetic (Julia mod, rewind, CAR affine, sky2pix roundings, taps)
//   pxl_elementwise.h    pix2sky / sky2pix streams            pxl_unwrap.h     unwind! scan, rewind!
//   pxl_maps.h           posmap, pixareamap                   pxl_tan.h        Gnomonic evaluators
//   pxl_reproject.h      tables, gather + register-staged     pxl_reproject_dma.h  the LDS-DMA kernel (fast path)
//   pxl_sample.h         CAR<->TAN reprojection, sampler      pxl_misc.h       FITS staging, synthetic data
// This file keeps the error plumbing and the extern "C" entry points.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared   (csrc/Makefile)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <mutex>
#include <vector>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <type_traits>

#include "pxl_device.h"

using namespace pxl;

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail(PXL_EHIP, "%s: %s", #expr, hipGetErrorString(e_));  \
    } while (0)

static int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(PXL_EHIP, "launch of %s failed: %s", what, hipGetErrorString(e));
    return PXL_OK;
}

static bool wcs_ok(const pxl_car_wcs* w) {
    if (!w) return false;
    for (int k = 0; k < 2; ++k)
        if (!std::isfinite(w->cdelt[k]) || !std::isfinite(w->crpix[k]) || !std::isfinite(w->crval[k]) ||
            w->cdelt[k] == 0.0)
            return false;
    return std::isfinite(w->unit) && w->unit != 0.0;
}

// Grid for 1-D streaming kernels: one contiguous chunk per block (measured best on MI355X: 69-73 % of HBM peak
// vs 57-67 % with a few thousand grid-striding blocks); grid-stride only past 2^20 blocks.
static inline unsigned stream_grid(int64_t work_items, int block) {
    static const int64_t cap = [] { const char* v = getenv("PXL_STREAM_BLOCKS"); return (v && *v) ? atoll(v) : (1LL << 20); }();
    int64_t nb = (work_items + block - 1) / block;
    if (nb < 1) nb = 1;
    if (nb > cap) nb = cap;
    return (unsigned)nb;
}

static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

#include "pxl_elementwise.h"
#include "pxl_unwrap.h"
#include "pxl_maps.h"
#include "pxl_fastmath.h"
#include "pxl_tan.h"
#include "pxl_reproject.h"
#include "pxl_reproject_dma.h"
#include "pxl_sample.h"
#include "pxl_misc.h"
#include "pxl_rccl.h"
#include "pxl_spread.h"

// ================================================================================================
// C ABI
// ================================================================================================
struct pxl_reproject_plan {
    pxl_car_wcs win, wout;
    int64_t nx, ny, nc, src_row0, src_nrows;
    int64_t nxo, nyo, dst_row0, dst_nrows;
    int periodic;
    int device;
    // device tables
    int32_t* xi0; double* xfx; int32_t* yj0; double* yfy;
    void* table_mem;
    // host copy of the row table (cells only), same arithmetic as the device
    int32_t* h_yj0;
    // launch configuration
    int variant;       // 0 auto, 1 gather, 2 staged
    int pairs;         // lane width of the register-staged kernel: 1 or 2 (x2 output columns per lane)
    int pairs_dma;     // lane width of the LDS-DMA kernel: 1, 2 or 4
    int seg_dma;
    int pairs_dma32;   // the same for Float32 storage (4 elements per lane per access)
    int seg_dma32;
    bool dma32_ok;
    int dypos;
    int rh;            // tile height for Float64 maps (Float32 launches use rh32)
    int rh32;
    int seg;
    int dxpos;
    int flags;
    int ns, pf;
    int ring_kb;       // LDS a wave's ring may take (KiB); the ring is halved until it fits
    int min_tiles;     // the tile height is halved (down to 4 rows) while a launch has fewer tiles than this
    int nt;            // non-temporal stores (LDS-DMA kernel, full tiles)
    int64_t xchunk;
    double* zero_page;
    bool staged_ok;
    bool vec_load;
    bool tables_built;
    // sharded step: the halo exchange runs on a stream of its own, fenced by two events
    hipStream_t comm_stream;
    hipEvent_t ev_ready, ev_halo;
    int64_t cov_have_lo, cov_have_hi, cov_lo, cov_hi;     // cached pxl_reproject_plan_rows_covered answer
};

// Scratch of one unwind! call, from the stream-ordered allocator (hipMallocAsync / hipFreeAsync): no host
// synchronisation.  The multi-pass fallback needs per-element scratch (5 B per value); the fused path only
// per-chunk entries.
struct UnwindWs {
    char* base;
    int8_t* c; int32_t* rloc; int32_t* bsum; int32_t* boff;     // multi-pass form
    int32_t* flag;                                              // [0],[1]: multi-pass verification; [2]: fused path failed
    unsigned long long* firstnan;                               // first NaN of each coordinate row
    int2* wsum; double2* wprev;                                 // fused form: per wave chunk
    UwLink* links; unsigned int* ticket; int64_t nlinks;        // one-pass form: one link per workgroup chunk
    int64_t nb, nw;
    int U;                                                      // points per wave chunk / 64
    bool multipass;                                             // per-element scratch of the multi-pass fallback present
};

// Batches up to this size fall back (half-period ties only) to ONE gated launch of the single-block form instead
// of the ten gated launches of the multi-pass form: ~25 us less per call, where that is most of the call.  The
// single block is slower when it does run (9 us per 4096 points and pass, then the serial recurrence), so longer
// batches keep the multi-pass form, whose fixed cost no longer matters there.
static const int64_t kUnwindBlockFallbackMax = 1LL << 22;


// The scratch comes from a library-owned stream-ordered pool that keeps what it has been given: with the default
// pool (release threshold 0) every call on a drained stream went back to the driver for its memory (~250 us).
static std::mutex g_pool_mu;
static hipMemPool_t g_pools[64] = {};
static hipMemPool_t unwind_pool() {
    std::mutex& mu = g_pool_mu;
    hipMemPool_t* pools = g_pools;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!pools[dev]) {
        hipMemPoolProps props = {};
        props.allocType = hipMemAllocationTypePinned;
        props.location.type = hipMemLocationTypeDevice;
        props.location.id = dev;
        hipMemPool_t p = nullptr;
        if (hipMemPoolCreate(&p, &props) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        uint64_t keep = ~0ull;
        (void)hipMemPoolSetAttribute(p, hipMemPoolAttrReleaseThreshold, &keep);
        pools[dev] = p;
    }
    return pools[dev];
}

static int unwind_ws_alloc(int64_t n, int nrow, hipStream_t st, UnwindWs* w) {
    w->nb = (n + PXL_SCAN_BLOCK - 1) / PXL_SCAN_BLOCK;
    // wave chunks of 64*U points: long enough to amortise the per-chunk work, short enough to fill the GPU
    int64_t U = n / (64 * 4096);
    U = U < 4 ? 4 : (U > 32 ? 32 : U);
    U &= ~(int64_t)(PXL_UW_G - 1);
    w->U = (int)U;
    w->nw = (n + 64 * U - 1) / (64 * U);
    if (w->nb > 0x7fffffffLL || w->nw > 0x7fffffffLL || n > 0x7fffffffLL) return fail(PXL_EINVAL, "unwind: batch too long (2^31 points)");
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    w->multipass = n > kUnwindBlockFallbackMax;
    const size_t bytes_c = w->multipass ? up((size_t)nrow * n) : 0, bytes_r = w->multipass ? up((size_t)4 * nrow * n) : 0,
                 bytes_b = w->multipass ? up((size_t)4 * nrow * w->nb) : 0;
    const size_t bytes_ws = up((size_t)w->nw * sizeof(int2)), bytes_wp = up((size_t)w->nw * sizeof(double2));
    w->nlinks = (n + PXL_UW1_CHUNK - 1) / PXL_UW1_CHUNK;
    const size_t bytes_ln = up((size_t)w->nlinks * sizeof(UwLink) + 16);
    const size_t total = bytes_c + bytes_r + 2 * bytes_b + bytes_ws + bytes_wp + bytes_ln + 256;
    w->base = nullptr;
    hipMemPool_t pool = unwind_pool();
    if (pool) HIP_TRY(hipMallocFromPoolAsync((void**)&w->base, total, pool, st));
    else HIP_TRY(hipMallocAsync((void**)&w->base, total, st));
    char* p = w->base;
    w->c = (int8_t*)p; p += bytes_c;
    w->rloc = (int32_t*)p; p += bytes_r;
    w->bsum = (int32_t*)p; p += bytes_b;
    w->boff = (int32_t*)p; p += bytes_b;
    w->wsum = (int2*)p; p += bytes_ws;
    w->wprev = (double2*)p; p += bytes_wp;
    w->links = (UwLink*)p; w->ticket = (unsigned int*)(p + (size_t)w->nlinks * sizeof(UwLink)); p += bytes_ln;
    w->flag = (int32_t*)p;
    w->firstnan = (unsigned long long*)(p + 16);
    if (hipMemsetAsync(w->flag, 0, 32, st) != hipSuccess) {      // flags and the (complemented) first-NaN indices
        (void)hipFreeAsync(w->base, st);
        return fail(PXL_EHIP, "unwind: hipMemsetAsync failed");
    }
    return PXL_OK;
}

static int unwind_ws_free(UnwindWs* w, hipStream_t st, int rc) {
    hipError_t e = hipFreeAsync(w->base, st);
    if (e != hipSuccess && rc == PXL_OK) rc = fail(PXL_EHIP, "unwind: hipFreeAsync: %s", hipGetErrorString(e));
    return rc;
}

// Fused form (pxl_unwrap.h): sums -> scan -> verify + write.  `inplace`: out aliases the input, so verify without
// storing first and store in a second launch that the device skips if the verification failed.
template <class SRC>
static int unwind_fused(const SRC& src, typename SRC::raw_t* out, int64_t n, bool inplace, UnwindWs& w, hipStream_t st) {
    const dim3 grid((unsigned)w.nw), block(64);
    int32_t* fastflag = w.flag + 2;
    hipLaunchKernelGGL((k_unwind_sums<SRC>), grid, block, 0, st, src, n, w.U, w.wsum, w.wprev, w.firstnan);
    hipLaunchKernelGGL(k_scan_wsums, dim3(1), dim3(1024), 0, st, w.nw, w.wsum);
    if (inplace) {
        hipLaunchKernelGGL((k_unwind_apply<SRC, false>), grid, block, 0, st, src, out, n, w.U, (const int2*)w.wsum,
                           (const double2*)w.wprev, (const unsigned long long*)w.firstnan, fastflag, (const int32_t*)nullptr);
        hipLaunchKernelGGL((k_unwind_apply<SRC, true>), grid, block, 0, st, src, out, n, w.U, (const int2*)w.wsum,
                           (const double2*)w.wprev, (const unsigned long long*)w.firstnan, fastflag, (const int32_t*)fastflag);
    } else {
        hipLaunchKernelGGL((k_unwind_apply<SRC, true>), grid, block, 0, st, src, out, n, w.U, (const int2*)w.wsum,
                           (const double2*)w.wprev, (const unsigned long long*)w.firstnan, fastflag, (const int32_t*)nullptr);
    }
    return check_launch("k_unwind fused");
}

// One-pass form (k_unwind_onepass: decoupled look-back; out-of-place calls only).  The links and the ticket are zeroed on the
// stream before the launch; the failure flag is the fused path's, so the same fallbacks follow.
template <class SRC>
static int unwind_onepass(const SRC& src, typename SRC::raw_t* out, int64_t n, UnwindWs& w, hipStream_t st) {
    if (w.nlinks > 0x7fffffffLL) return fail(PXL_EINVAL, "unwind: batch too long");
    hipError_t e = hipMemsetAsync(w.links, 0, (size_t)w.nlinks * sizeof(UwLink) + 16, st);
    if (e != hipSuccess) return fail(PXL_EHIP, "unwind: hipMemsetAsync: %s", hipGetErrorString(e));
    const int64_t nwg = (n + PXL_UW1_CHUNK - 1) / PXL_UW1_CHUNK;
    hipLaunchKernelGGL((k_unwind_onepass<SRC, PXL_UW1_U, PXL_UW1_UL>), dim3((unsigned)nwg), dim3(64 * PXL_UW1_WAVES), 0, st, src, out, n, w.links, w.ticket, w.flag + 2);
    return check_launch("k_unwind_onepass");
}

// Multi-pass form on a buffer that already holds m = rewind(.) - ref (see k_unwrap_* in pxl_unwrap.h).  With
// `gate` every launch is skipped on the device unless *gate != 0 (the fused form failed its verification).
static int unwind_multipass(int64_t n, int nrow, double* sky, double period, double ref, UnwindWs& w, const int32_t* gate,
                            hipStream_t st) {
    int rc = PXL_OK;
    // as a fallback these launches almost always exit at the gate: a small grid keeps that to a few microseconds each
    // (390k empty blocks cost ~85 us per launch), and the kernels grid-stride when they do run
    const unsigned cap = gate ? 2048u : 0xffffffffu;
    const unsigned g = std::min(stream_grid(n, 256), cap);
    const int64_t nb = w.nb;
    hipLaunchKernelGGL(k_unwrap_incr, dim3(g), dim3(256), 0, st, n, nrow, (const double*)sky, period, w.c, gate);
    for (int pass = 0; pass < 2 && rc == PXL_OK; ++pass) {
        const int32_t* pg = pass == 0 ? gate : w.flag;          // pass 2 runs on the device only if pass 1 flagged
        hipLaunchKernelGGL((k_scan_local<int8_t>), dim3(std::min((unsigned)nb, cap), nrow), dim3(256), 0, st, n, (const int8_t*)w.c, w.rloc, w.bsum, nb, pg);
        hipLaunchKernelGGL(k_scan_bsums, dim3(nrow), dim3(1024), 0, st, nb, (const int32_t*)w.bsum, w.boff, pg);
        hipLaunchKernelGGL(k_unwrap_verify, dim3(g), dim3(256), 0, st, n, nrow, (const double*)sky, period, w.c,
                           (const int32_t*)w.rloc, (const int32_t*)w.boff, nb, w.flag + pass, pg);
        rc = check_launch("k_unwrap scan/verify");
    }
    if (rc == PXL_OK) {
        hipLaunchKernelGGL(k_unwrap_apply, dim3(g), dim3(256), 0, st, n, nrow, sky, period, ref, (const int32_t*)w.rloc,
                           (const int32_t*)w.boff, nb, (const int32_t*)w.flag, gate);
        hipLaunchKernelGGL(k_unwind_rows, dim3(nrow), dim3(64), 0, st, n, nrow, sky, period, ref, 1, (const int32_t*)w.flag);
        rc = check_launch("k_unwrap_apply");
    }
    return rc;
}


extern "C" {

int pxl_version(void) { return PXL_VERSION; }

size_t pxl_last_error(char* buf, size_t n) {
    size_t len = strlen(g_err);
    if (buf && n) {
        size_t m = len < n - 1 ? len : n - 1;
        memcpy(buf, g_err, m);
        buf[m] = 0;
    }
    return len;
}

int pxl_release_scratch(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return fail(PXL_ENODEV, "release_scratch: no current device");
    std::lock_guard<std::mutex> lock(g_pool_mu);
    if (g_pools[dev] && hipMemPoolTrimTo(g_pools[dev], 0) != hipSuccess) return fail(PXL_EHIP, "release_scratch: hipMemPoolTrimTo failed");
    return PXL_OK;
}

// ---- placement probe (pxl_spread.h): eight store fronts, four in each of two windows
int pxl_mem_probe_pair(void* a, void* b, size_t window_bytes, int reps, float* us, void* stream) {
    if (!a || !b || !us) return fail(PXL_EINVAL, "mem_probe_pair: null argument");
    if (window_bytes < (64u << 20) || (window_bytes & 63) != 0 || (((uintptr_t)a | (uintptr_t)b) & 15) != 0)
        return fail(PXL_EINVAL, "mem_probe_pair: windows of at least 64 MiB, a multiple of 64 bytes, 16-byte aligned");
    if (reps < 1 || reps > 99) return fail(PXL_EINVAL, "mem_probe_pair: 1..99 repetitions");
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    {
        hipError_t ec = hipEventCreate(&e1);
        if (ec != hipSuccess) { (void)hipEventDestroy(e0); return fail(PXL_EHIP, "hipEventCreate(&e1): %s", hipGetErrorString(ec)); }
    }
    std::vector<float> t(reps);
    int rc = PXL_OK;
    for (int r = -1; r < reps && rc == PXL_OK; ++r) {
        hipError_t e = hipEventRecord(e0, st);
        hipLaunchKernelGGL(k_spread_probe, dim3(8 * 256 * 2), dim3(256), 0, st, (char*)a, (char*)b, window_bytes / 4);
        if (e == hipSuccess) e = hipGetLastError();          // a refused launch would otherwise time an empty interval
        if (e == hipSuccess) e = hipEventRecord(e1, st);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e != hipSuccess) rc = fail(PXL_EHIP, "mem_probe_pair: %s", hipGetErrorString(e));
        else if (r >= 0) t[r] = ms * 1000.f;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc) return rc;
    std::sort(t.begin(), t.end());
    *us = t[reps / 2];
    return PXL_OK;
}

#include "pxl_place.h"

int pxl_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(PXL_ENODEV, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

int pxl_pix2sky_car_f64(const pxl_car_wcs* wcs, int64_t n, const double* pix, double* sky, int wrap_mode,
                        void* stream) {
    if (!wcs_ok(wcs)) return fail(PXL_EINVAL, "pix2sky: invalid WCS");
    if (n < 0 || (n > 0 && (!pix || !sky))) return fail(PXL_EINVAL, "pix2sky: null buffer or negative n");
    if (wrap_mode < PXL_WRAP_NONE || wrap_mode > PXL_WRAP_UNWIND) return fail(PXL_EINVAL, "pix2sky: bad wrap_mode %d", wrap_mode);
    if ((((uintptr_t)pix | (uintptr_t)sky) & 15) != 0) return fail(PXL_EINVAL, "pix2sky: 2xN buffers must be 16-byte aligned");
    if (n == 0) return PXL_OK;
    hipStream_t st = (hipStream_t)stream;
    CarAffine c = car_affine(*wcs);
    const int mode = wrap_mode == PXL_WRAP_REWIND ? 1 : (wrap_mode == PXL_WRAP_UNWIND ? 2 : 0);
    if (wrap_mode == PXL_WRAP_NONE) {          // affine only: one point per lane per trip
        hipLaunchKernelGGL((k_pix2sky_pairs<1>), dim3(stream_grid(n, 256)), dim3(256), 0, st, c, n, (const double2*)pix,
                           (double2*)sky, mode, (const int32_t*)nullptr);
        return check_launch("k_pix2sky_pairs");
    }
    const dim3 pgrid(stream_grid((n + 1) / 2, 256));
    if (wrap_mode != PXL_WRAP_UNWIND) {
        hipLaunchKernelGGL((k_pix2sky_pairs<2>), pgrid, dim3(256), 0, st, c, n, (const double2*)pix, (double2*)sky, mode,
                           (const int32_t*)nullptr);
        return check_launch("k_pix2sky_pairs");
    }
    // in-place (pix == sky) is fine, a partial overlap is not: every unwind form reads inputs other waves / rounds
    // have not consumed yet while it stores (checked before ANY launch, the single-block form included)
    const uintptr_t pa = (uintptr_t)pix, sa = (uintptr_t)sky, bytes = (uintptr_t)n * 16;
    if (pa != sa && pa < sa + bytes && sa < pa + bytes) return fail(PXL_EINVAL, "pix2sky: pix and sky may alias exactly or not at all");
    if (n <= PXL_UWB_MAX) {       // small batch: everything in one launch of one block
        UwSrcPix2 src{c, (const double2*)pix, PXL_TWOPI_D, 0.0, 1.0 / PXL_TWOPI_D};
        hipLaunchKernelGGL((k_unwind_block<UwSrcPix2>), dim3(1), dim3(1024), 0, st, src, (double2*)sky, n, (const int32_t*)nullptr);
        return check_launch("k_unwind_block");
    }
    // safe=true on a long batch: fused rewind + verified scan; the multi-pass form only if its check fails
    UnwindWs w;
    int rc = unwind_ws_alloc(n, 2, st, &w);
    if (rc) return rc;
    UwSrcPix2 src{c, (const double2*)pix, PXL_TWOPI_D, 0.0, 1.0 / PXL_TWOPI_D};
    // out of place: one pass (input read once, the exact rewind evaluated once); in place: sums -> scan -> verify -> store
    if (pa != sa && n < PXL_UW_ONEPASS_MAX && env_int("PXL_UNWIND_ONEPASS", 1)) rc = unwind_onepass(src, (double2*)sky, n, w, st);
    else rc = unwind_fused(src, (double2*)sky, n, pa == sa, w, st);
    if (rc == PXL_OK) {
        const int32_t* failed = w.flag + 2;
        if (!w.multipass) {
            hipLaunchKernelGGL((k_unwind_block<UwSrcPix2>), dim3(1), dim3(1024), 0, st, src, (double2*)sky, n, failed);
            rc = check_launch("k_unwind_block");
        } else {
            hipLaunchKernelGGL((k_pix2sky_pairs<2>), dim3(std::min(pgrid.x, 2048u)), dim3(256), 0, st, c, n, (const double2*)pix, (double2*)sky, 2, failed);
            rc = check_launch("k_pix2sky_pairs");
            if (rc == PXL_OK) rc = unwind_multipass(n, 2, sky, PXL_TWOPI_D, 0.0, w, failed, st);
        }
    }
    return unwind_ws_free(&w, st, rc);
}

int pxl_rewind_f64(double* a, int64_t n, double period, double ref_angle, void* stream) {
    if (n < 0 || (n > 0 && !a)) return fail(PXL_EINVAL, "rewind: null buffer or negative n");
    if (!(period > 0.0) || !std::isfinite(period) || !std::isfinite(ref_angle)) return fail(PXL_EINVAL, "rewind: period must be positive and finite");
    if (n == 0) return PXL_OK;
    hipLaunchKernelGGL(k_rewind, dim3(stream_grid((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, n, a, period, ref_angle, 0,
                       (const int32_t*)nullptr);
    return check_launch("k_rewind");
}

int pxl_unwind_f64(double* a, int64_t n, int nrow, double period, double ref_angle, void* stream) {
    if (n < 0 || (n > 0 && !a)) return fail(PXL_EINVAL, "unwind: null buffer or negative n");
    if (nrow != 1 && nrow != 2) return fail(PXL_EINVAL, "unwind: nrow must be 1 (vector) or 2 (2xN batch)");
    if (!(period > 0.0) || !std::isfinite(period) || !std::isfinite(ref_angle)) return fail(PXL_EINVAL, "unwind: period must be positive and finite");
    if (nrow == 2 && ((uintptr_t)a & 15) != 0) return fail(PXL_EINVAL, "unwind: 2xN buffer must be 16-byte aligned");
    if (n == 0) return PXL_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 rgrid(stream_grid(((int64_t)nrow * n + 3) / 4, 256));
    if (n <= PXL_UWB_MAX) {
        if (nrow == 2) {
            UwSrcAng2 src{(const double2*)a, period, ref_angle, 1.0 / period};
            hipLaunchKernelGGL((k_unwind_block<UwSrcAng2>), dim3(1), dim3(1024), 0, st, src, (double2*)a, n, (const int32_t*)nullptr);
        } else {
            UwSrcAng1 src{(const double*)a, period, ref_angle, 1.0 / period};
            hipLaunchKernelGGL((k_unwind_block<UwSrcAng1>), dim3(1), dim3(1024), 0, st, src, a, n, (const int32_t*)nullptr);
        }
        return check_launch("k_unwind_block");
    }
    UnwindWs w;
    int rc = unwind_ws_alloc(n, nrow, st, &w);
    if (rc) return rc;
    const int32_t* failed = w.flag + 2;
    if (nrow == 2) {
        UwSrcAng2 src{(const double2*)a, period, ref_angle, 1.0 / period};
        rc = unwind_fused(src, (double2*)a, n, true, w, st);
        if (rc == PXL_OK && !w.multipass)
            hipLaunchKernelGGL((k_unwind_block<UwSrcAng2>), dim3(1), dim3(1024), 0, st, src, (double2*)a, n, failed);
    } else {
        UwSrcAng1 src{(const double*)a, period, ref_angle, 1.0 / period};
        rc = unwind_fused(src, a, n, true, w, st);
        if (rc == PXL_OK && !w.multipass)
            hipLaunchKernelGGL((k_unwind_block<UwSrcAng1>), dim3(1), dim3(1024), 0, st, src, a, n, failed);
    }
    if (rc == PXL_OK && !w.multipass) rc = check_launch("k_unwind_block");
    if (rc == PXL_OK && w.multipass) {
        hipLaunchKernelGGL(k_rewind, dim3(std::min(rgrid.x, 2048u)), dim3(256), 0, st, (int64_t)nrow * n, a, period, ref_angle, 1, failed);
        rc = check_launch("k_rewind");
        if (rc == PXL_OK) rc = unwind_multipass(n, nrow, a, period, ref_angle, w, failed, st);
    }
    return unwind_ws_free(&w, st, rc);
}

int pxl_pix2sky_car_soa_f64(const pxl_car_wcs* wcs, int64_t n, const double* ipix, const double* jpix,
                            double* ra, double* dec, int safe, void* stream) {
    if (!wcs_ok(wcs)) return fail(PXL_EINVAL, "pix2sky_soa: invalid WCS");
    if (n < 0 || (n > 0 && (!ipix || !jpix || !ra || !dec))) return fail(PXL_EINVAL, "pix2sky_soa: null buffer or negative n");
    if (n == 0) return PXL_OK;
    const int vec = ((((uintptr_t)ipix | (uintptr_t)jpix | (uintptr_t)ra | (uintptr_t)dec) & 15) == 0) ? 1 : 0;
    hipLaunchKernelGGL(k_pix2sky_soa, dim3(stream_grid((n + 1) / 2, 256)), dim3(256), 0, (hipStream_t)stream,
                       car_affine(*wcs), n, ipix, jpix, ra, dec, safe ? 1 : 0, vec);
    return check_launch("k_pix2sky_soa");
}

int pxl_sky2pix_car_f64(const pxl_car_wcs* wcs, const int64_t shape[2], int64_t n, const double* sky,
                        double* pix, int safe, int form, void* stream) {
    if (!wcs_ok(wcs) || !shape) return fail(PXL_EINVAL, "sky2pix: invalid WCS/shape");
    if (n < 0 || (n > 0 && (!pix || !sky))) return fail(PXL_EINVAL, "sky2pix: null buffer or negative n");
    if (form < 0 || form > 2) return fail(PXL_EINVAL, "sky2pix: bad form %d", form);
    if ((((uintptr_t)pix | (uintptr_t)sky) & 15) != 0) return fail(PXL_EINVAL, "sky2pix: 2xN buffers must be 16-byte aligned");
    if (n == 0) return PXL_OK;
    Sky2Pix s = sky2pix_setup(*wcs, shape[0], shape[1], safe ? 1 : 0, form);
    if (safe)
        hipLaunchKernelGGL((k_sky2pix_pairs<2>), dim3(stream_grid((n + 1) / 2, 256)), dim3(256), 0,
                           (hipStream_t)stream, s, n, (const double2*)sky, (double2*)pix);
    else
        hipLaunchKernelGGL((k_sky2pix_pairs<1>), dim3(stream_grid(n, 256)), dim3(256), 0,
                           (hipStream_t)stream, s, n, (const double2*)sky, (double2*)pix);
    return check_launch("k_sky2pix_pairs");
}

int pxl_sky2pix_car_soa_f64(const pxl_car_wcs* wcs, const int64_t shape[2], int64_t n, const double* ra,
                            const double* dec, double* ipix, double* jpix, int safe, int form, void* stream) {
    if (!wcs_ok(wcs) || !shape) return fail(PXL_EINVAL, "sky2pix_soa: invalid WCS/shape");
    if (n < 0 || (n > 0 && (!ipix || !jpix || !ra || !dec))) return fail(PXL_EINVAL, "sky2pix_soa: null buffer or negative n");
    if (form < 0 || form > 2) return fail(PXL_EINVAL, "sky2pix_soa: bad form %d", form);
    if (n == 0) return PXL_OK;
    Sky2Pix s = sky2pix_setup(*wcs, shape[0], shape[1], safe ? 1 : 0, form);
    const int vec = ((((uintptr_t)ipix | (uintptr_t)jpix | (uintptr_t)ra | (uintptr_t)dec) & 15) == 0) ? 1 : 0;
    hipLaunchKernelGGL(k_sky2pix_soa, dim3(stream_grid((n + 1) / 2, 256)), dim3(256), 0, (hipStream_t)stream, s, n, ra,
                       dec, ipix, jpix, vec);
    return check_launch("k_sky2pix_soa");
}

static int check_rows(const char* who, const int64_t shape[2], int64_t row0, int64_t nrows) {
    if (!shape || shape[0] < 1 || shape[1] < 1) return fail(PXL_EINVAL, "%s: bad shape", who);
    if (row0 < 0 || nrows < 0 || row0 + nrows > shape[1])
        return fail(PXL_EINVAL, "%s: rows [%lld, %lld) outside the map (ny=%lld)", who, (long long)row0,
                    (long long)(row0 + nrows), (long long)shape[1]);
    return PXL_OK;
}

int pxl_posmap_car_f64(const pxl_car_wcs* wcs, const int64_t shape[2], int64_t row0, int64_t nrows,
                       double* ra, double* dec, int safe, void* stream) {
    if (!wcs_ok(wcs)) return fail(PXL_EINVAL, "posmap: invalid WCS");
    int rc = check_rows("posmap", shape, row0, nrows);
    if (rc) return rc;
    if (nrows == 0) return PXL_OK;
    if (!ra || !dec) return fail(PXL_EINVAL, "posmap: null output");
    if (nrows > 65535LL * PXL_POS_ROWS) return fail(PXL_EINVAL, "posmap: more than %lld rows per call", 65535LL * PXL_POS_ROWS);
    const int64_t nych = (nrows + PXL_POS_ROWS - 1) / PXL_POS_ROWS;
    int fronts = env_int("PXL_POSMAP_FRONTS", 8);
    if (fronts < 1 || nych < 16 * fronts) fronts = 1;
    int64_t per = (nych + fronts - 1) / fronts;
    if (per * fronts > 65535) { fronts = 1; per = nych; }      // rounding up to a multiple of `fronts` must not pass the grid.y limit
    dim3 grid((unsigned)(((shape[0] + 1) / 2 + 255) / 256), (unsigned)(per * fronts));
    hipLaunchKernelGGL(k_posmap_car, grid, dim3(256), 0, (hipStream_t)stream,
                       car_affine(*wcs), shape[0], row0, nrows, ra, dec, safe ? 1 : 0, fronts);
    return check_launch("k_posmap_car");
}

int pxl_pixareamap_car_f64(const pxl_car_wcs* wcs, const int64_t shape[2], int64_t row0, int64_t nrows,
                           double* area, void* stream) {
    if (!wcs_ok(wcs)) return fail(PXL_EINVAL, "pixareamap: invalid WCS");
    int rc = check_rows("pixareamap", shape, row0, nrows);
    if (rc) return rc;
    if (nrows == 0) return PXL_OK;
    if (!area) return fail(PXL_EINVAL, "pixareamap: null output");
    if ((shape[0] & 1) == 0 && ((uintptr_t)area & 15) == 0 && env_int("PXL_AREA_ROWS", 0) == 0) {
        // one contiguous chunk of the map per block (pairs of pixels, 16-byte stores)
        const int64_t total = shape[0] / 2 * nrows;
        const int64_t nchunks = (total + PXL_AREA_CHUNK - 1) / PXL_AREA_CHUNK;
        int fronts = env_int("PXL_AREA_FRONTS", 8);
        if (fronts < 1 || nchunks < 64 * fronts) fronts = 1;
        const int64_t per = (nchunks + fronts - 1) / fronts;
        hipLaunchKernelGGL(k_pixareamap_chunks, dim3((unsigned)(per * fronts)), dim3(256), 0,
                           (hipStream_t)stream, car_affine(*wcs), shape[0], row0, nrows, area, fronts);
        return check_launch("k_pixareamap_chunks");
    }
    // odd nx / unaligned map: one row per blockIdx.y (<= 65535 per launch)
    for (int64_t r = 0; r < nrows; r += 65535) {
        int64_t nr = (nrows - r < 65535) ? nrows - r : 65535;
        unsigned gx = (unsigned)std::max<int64_t>(1, ((shape[0] + 1) / 2 + 2047) / 2048);   // ~8 pairs per lane
        hipLaunchKernelGGL(k_pixareamap_car, dim3(gx, (unsigned)nr), dim3(256), 0, (hipStream_t)stream,
                           car_affine(*wcs), shape[0], row0 + r, nr, area + r * shape[0]);
        int rc2 = check_launch("k_pixareamap_car");
        if (rc2) return rc2;
    }
    return PXL_OK;
}

int pxl_sky2pix_tan_f64(const pxl_car_wcs* wcs, int64_t n, const double* ra, const double* dec, double* ipix,
                        double* jpix, void* stream) {
    if (!wcs_ok(wcs)) return fail(PXL_EINVAL, "sky2pix_tan: invalid WCS");
    if (n < 0 || (n > 0 && (!ipix || !jpix || !ra || !dec))) return fail(PXL_EINVAL, "sky2pix_tan: null buffer or negative n");
    if (n == 0) return PXL_OK;
    const bool vec = ((((uintptr_t)ra | (uintptr_t)dec | (uintptr_t)ipix | (uintptr_t)jpix) & 15) == 0);
    const dim3 grid(stream_grid((n + 1) / 2, 256));
    if (vec) hipLaunchKernelGGL((k_tan_points<true, false>), grid, dim3(256), 0, (hipStream_t)stream, tan_setup(*wcs), n, ra, dec, ipix, jpix);
    else     hipLaunchKernelGGL((k_tan_points<false, false>), grid, dim3(256), 0, (hipStream_t)stream, tan_setup(*wcs), n, ra, dec, ipix, jpix);
    return check_launch("k_tan_points (sky2pix)");
}

int pxl_pix2sky_tan_f64(const pxl_car_wcs* wcs, int64_t n, const double* ipix, const double* jpix, double* ra,
                        double* dec, void* stream) {
    if (!wcs_ok(wcs)) return fail(PXL_EINVAL, "pix2sky_tan: invalid WCS");
    if (n < 0 || (n > 0 && (!ipix || !jpix || !ra || !dec))) return fail(PXL_EINVAL, "pix2sky_tan: null buffer or negative n");
    if (n == 0) return PXL_OK;
    const bool vec = ((((uintptr_t)ra | (uintptr_t)dec | (uintptr_t)ipix | (uintptr_t)jpix) & 15) == 0);
    const dim3 grid(stream_grid((n + 1) / 2, 256));
    if (vec) hipLaunchKernelGGL((k_tan_points<true, true>), grid, dim3(256), 0, (hipStream_t)stream, tan_setup(*wcs), n, ipix, jpix, ra, dec);
    else     hipLaunchKernelGGL((k_tan_points<false, true>), grid, dim3(256), 0, (hipStream_t)stream, tan_setup(*wcs), n, ipix, jpix, ra, dec);
    return check_launch("k_tan_points (pix2sky)");
}

int pxl_posmap_tan_f64(const pxl_car_wcs* wcs, const int64_t shape[2], int64_t row0, int64_t nrows, double* ra,
                       double* dec, void* stream) {
    if (!wcs_ok(wcs)) return fail(PXL_EINVAL, "posmap_tan: invalid WCS");
    int rc = check_rows("posmap_tan", shape, row0, nrows);
    if (rc) return rc;
    if (nrows == 0) return PXL_OK;
    if (!ra || !dec) return fail(PXL_EINVAL, "posmap_tan: null output");
    const int64_t nchunk = (shape[0] + 511) / 512;                 // 512 RA pixels per block
    const int64_t nrb = (nrows + PXL_TAN_ROWS - 1) / PXL_TAN_ROWS;   // PXL_TAN_ROWS rows per block
    if (nchunk * nrb > 0x7fffffffLL) return fail(PXL_EINVAL, "posmap_tan: map too large for one launch");
    const bool vec = (shape[0] % 2 == 0) && ((((uintptr_t)ra | (uintptr_t)dec) & 15) == 0);
    // maps of at least one tile: the grid form (anchors + closed-form differences + interpolation, pxl_tan.h); PXL_TAN_GRID=0 or a
    // small map: the per-pixel evaluation
    // The grid form pays when a 128-column tile spans at most ~2 degrees (pixels up to ~1 arcmin: beyond that the degree-8 interpolant
    // misses its 2^-55 rad check and the rows fall back to the per-pixel path after paying for the lattice) and the patch centre is
    // not next to a pole (there most rows fail the small-angle preconditions)
    const TanParams tp0 = tan_setup(*wcs);
    const bool grid_pays = fabs(tp0.uos) * PXL_TG_W <= 0.04 && fabs(tp0.cd0) >= 0.3;
    if (env_int("PXL_TAN_GRID", grid_pays ? 1 : 0) && shape[0] >= PXL_TG_W && nrows >= 8) {
        const int64_t ntx = (shape[0] + PXL_TG_W - 1) / PXL_TG_W, nty = (nrows + PXL_TG_ROWS - 1) / PXL_TG_ROWS;
        int fronts = env_int("PXL_POSMAP_FRONTS", 8);
        if (fronts < 1 || nty < 4 * fronts) fronts = 1;
        const int64_t per = (nty + fronts - 1) / fronts, nbx = (ntx + 3) / 4;
        const int64_t nblk = per * fronts * nbx;
        if (nblk <= 0x7fffffffLL) {
            if (vec) hipLaunchKernelGGL((k_posmap_tan_grid<true>), dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, tp0, shape[0], row0, nrows, ntx, per, fronts, ra, dec);
            else     hipLaunchKernelGGL((k_posmap_tan_grid<false>), dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, tp0, shape[0], row0, nrows, ntx, per, fronts, ra, dec);
            return check_launch("k_posmap_tan_grid");
        }
    }
    const dim3 grid((unsigned)(nchunk * nrb));
    if (vec) hipLaunchKernelGGL((k_posmap_tan<true>), grid, dim3(256), 0, (hipStream_t)stream, tan_setup(*wcs), shape[0], row0, nrows, nchunk, ra, dec);
    else     hipLaunchKernelGGL((k_posmap_tan<false>), grid, dim3(256), 0, (hipStream_t)stream, tan_setup(*wcs), shape[0], row0, nrows, nchunk, ra, dec);
    return check_launch("k_posmap_tan");
}

// ---- reprojection plan -------------------------------------------------------------------------

int pxl_reproject_plan_create(const pxl_car_wcs* wcs_in, const int64_t shape_in[3], int64_t src_row0,
                              int64_t src_nrows, const pxl_car_wcs* wcs_out, const int64_t shape_out[2],
                              int64_t dst_row0, int64_t dst_nrows, pxl_reproject_plan** out) {
    if (!out) return fail(PXL_EINVAL, "plan_create: null plan pointer");
    *out = nullptr;
    if (!wcs_ok(wcs_in) || !wcs_ok(wcs_out)) return fail(PXL_EINVAL, "plan_create: invalid WCS");
    if (!shape_in || !shape_out) return fail(PXL_EINVAL, "plan_create: null shape");
    if (shape_in[0] < 1 || shape_in[1] < 1 || shape_in[2] < 1 || shape_out[0] < 1 || shape_out[1] < 1)
        return fail(PXL_EINVAL, "plan_create: shapes must be positive");
    // int32 cell tables and 32-bit byte offsets within a source row: nx * 8 must stay below 2^32
    if (shape_in[0] > 400000000 || shape_in[1] > 1000000000 || shape_out[0] > 1000000000 || shape_out[1] > 1000000000)
        return fail(PXL_EINVAL, "plan_create: axis too long (source RA axis <= 4e8, others <= 1e9 pixels)");
    if (src_row0 < 0 || src_nrows < 0 || src_row0 + src_nrows > shape_in[1])
        return fail(PXL_EINVAL, "plan_create: source window outside the map");
    if (dst_row0 < 0 || dst_nrows < 0 || dst_row0 + dst_nrows > shape_out[1])
        return fail(PXL_EINVAL, "plan_create: destination window outside the map");

    pxl_reproject_plan* pl = new (std::nothrow) pxl_reproject_plan();     // value-initialised: all members zero
    if (!pl) return fail(PXL_ENOMEM, "plan_create: host allocation failed");
    pl->cov_lo = 0; pl->cov_hi = -1;                                        // no cached interior yet
    pl->win = *wcs_in; pl->wout = *wcs_out;
    pl->nx = shape_in[0]; pl->ny = shape_in[1]; pl->nc = shape_in[2];
    pl->src_row0 = src_row0; pl->src_nrows = src_nrows;
    pl->nxo = shape_out[0]; pl->nyo = shape_out[1];
    pl->dst_row0 = dst_row0; pl->dst_nrows = dst_nrows;
    // full-circle test: same 1e-8 threshold as enmap_geom.jl:55
    pl->periodic = fabs((double)pl->nx * fabs(wcs_in->cdelt[0] * wcs_in->unit) - PXL_TWOPI_D) < 1e-8;
    pl->tables_built = false;
    pl->variant = env_int("PXL_REPROJECT_VARIANT", 0);

    hipError_t e = hipGetDevice(&pl->device);
    if (e != hipSuccess) { delete pl; return fail(PXL_ENODEV, "hipGetDevice: %s", hipGetErrorString(e)); }

    // tables: [xfx nxo doubles][yfy nyo doubles][xi0 nxo int32][yj0 nyo int32], 16-B aligned pieces
    size_t nxo = (size_t)pl->nxo, nyo = (size_t)pl->nyo;
    size_t off_xfx = 0;
    size_t off_yfy = off_xfx + ((nxo * 8 + 15) & ~(size_t)15);
    size_t off_xi0 = off_yfy + ((nyo * 8 + 15) & ~(size_t)15);
    size_t off_yj0 = off_xi0 + ((nxo * 4 + 15) & ~(size_t)15);
    size_t off_zero = off_yj0 + ((nyo * 4 + 15) & ~(size_t)15);
    size_t total = off_zero + 64;
    e = hipMalloc(&pl->table_mem, total);
    if (e != hipSuccess) { delete pl; return fail(PXL_ENOMEM, "plan_create: hipMalloc(%zu): %s", total, hipGetErrorString(e)); }
    char* base = (char*)pl->table_mem;
    e = hipMemset(base + off_zero, 0, 64);
    if (e != hipSuccess) { (void)hipFree(pl->table_mem); delete pl; return fail(PXL_EHIP, "plan_create: hipMemset: %s", hipGetErrorString(e)); }
    pl->zero_page = (double*)(base + off_zero);
    pl->xfx = (double*)(base + off_xfx); pl->yfy = (double*)(base + off_yfy);
    pl->xi0 = (int32_t*)(base + off_xi0); pl->yj0 = (int32_t*)(base + off_yj0);

    // host copy of the row cells (identical arithmetic: fmod/div/floor are exact or correctly rounded)
    pl->h_yj0 = new (std::nothrow) int32_t[nyo];
    if (!pl->h_yj0) { (void)hipFree(pl->table_mem); delete pl; return fail(PXL_ENOMEM, "plan_create: host allocation failed"); }
    CarAffine co = car_affine(pl->wout);
    Sky2Pix si = sky2pix_setup(pl->win, pl->nx, pl->ny, 1, PXL_FORM_DIV);
    for (int64_t j = 0; j < pl->nyo; ++j) {
        double fr;
        split_cell(s2p_y(si, p2s_dec(co, (double)(j + 1))), &pl->h_yj0[j], &fr);
    }

    // ---- choose the launch configuration from the RA scale (source columns per output column)
    double sx = fabs((pl->wout.cdelt[0] * pl->wout.unit) / (pl->win.cdelt[0] * pl->win.unit));
    pl->dxpos = ((pl->wout.cdelt[0] * pl->wout.unit) / (pl->win.cdelt[0] * pl->win.unit)) > 0 ? 1 : 0;
    pl->dypos = ((pl->wout.cdelt[1] * pl->wout.unit) / (pl->win.cdelt[1] * pl->win.unit)) > 0 ? 1 : 0;
    double sy = fabs((pl->wout.cdelt[1] * pl->wout.unit) / (pl->win.cdelt[1] * pl->win.unit));
    // tile height: 32 output rows when up-sampling in DEC, 16 when a tile consumes about as many source rows as it
    // writes (measured on five buffer placements of the 0.5-arcmin IQU map: 1.3-3.7 % faster than 32, 77.2 % at best;
    // the 2x refinement prefers 32 by 1.5 %)
    // Other scale factors (round 3, profiles/r03_tune_other3_*.txt): 4x refinement 8 rows (81.8 % against 75.2 % with 32; 16: 80.2 %),
    // 2x coarsening 4 rows (75.4 % against 73.2 % with 16)
    // Round 4, timed in BURSTS of one plan (a launch's stores leave the caches in a state that moves the next launch by 2-20 %, so
    // interleaving single launches of different variants misleads; profiles/r04_tune_nt_bursts.txt): with non-temporal stores the 2x
    // refinement prefers 16 rows as well (1.486 -> 1.453 ms = 80.3 % on placed maps, 1.726 -> 1.713 in a one-class placement; strips equal)
    pl->rh = env_int("PXL_REPROJECT_RH", sy >= 1.5 ? 4 : (sy > 0.3 ? 16 : 8));
    if (pl->rh < 1) pl->rh = 1;
    if (pl->rh > 64) pl->rh = 64;          // one lane per tile row holds the row-table entry
    pl->rh32 = env_int("PXL_REPROJECT_RH", 32);      // Float32 maps (4 pixels per lane) prefer 32 in both regimes
    if (pl->rh32 < 1) pl->rh32 = 1;
    if (pl->rh32 > 64) pl->rh32 = 64;
    pl->flags = env_int("PXL_REPROJECT_FLAGS", 0);
    // ring depth: 8 slots; 4 (twice the resident waves) when the whole source sits in the Infinity Cache, where the latency to hide
    // is short and occupancy wins (config 2, 4096x2049 -> 2x: 0.089 vs 0.103 ms; profiles/r03_tune_pf2_cfg2.txt).  Larger launches
    // were measured both ways: the IQU map and its strips lose 1.7-2 % with 4, the 1' single-plane map gains 1.9 %
    pl->ns = env_int("PXL_REPROJECT_NS", (double)pl->nx * (double)pl->ny * (double)pl->nc * 8.0 <= 128.0 * 1048576.0 ? 4 : 8);
    if (pl->ns < 4) pl->ns = 4;
    while (pl->ns & (pl->ns - 1)) pl->ns &= pl->ns - 1;       // power of two
    if (pl->ns > 64) pl->ns = 64;
    pl->xchunk = env_int("PXL_REPROJECT_XCHUNK", 0);
    // prefetch distance in OUTPUT rows.  What hides the latency is the number of SOURCE rows in flight, and the ring allows ns - 2 of
    // them beyond the two being read: at 2x refinement 3 output rows ahead were 1.5 source rows (1.58 ms), 12 are 6 (1.51 ms;
    // profiles/r03_tune_pf_cfg3.txt); at equal resolution 2 ... 6 measure the same
    {
        int want_pf = (int)ceil((pl->ns - 2) / (sy > 0.125 ? sy : 0.125));
        pl->pf = env_int("PXL_REPROJECT_PF", want_pf < 3 ? 3 : want_pf);
    }
    pl->ring_kb = env_int("PXL_REPROJECT_RING_KB", 17);
    // 8192 tiles = 3.5 waves per resident slot.  The floor used to be 65 536 (round 1: "a few rounds of 16 waves per CU"), which cut
    // the tiles of every launch below ~0.5 GB down to 4-8 rows -- each tile then pays its column setup for a handful of rows: a 1/4
    // strip of the 2x refinement 0.547 -> 0.407 ms (53 -> 72 %), a 1/8 strip 0.295 -> 0.227 ms, the 1' same-resolution map 0.634 ->
    // 0.620 ms, config 2 0.090 -> 0.082 ms; the IQU strips have more tiles than either floor (profiles/r03_tune_mintiles*.txt)
    pl->min_tiles = env_int("PXL_REPROJECT_MIN_TILES", 8192);
    // non-temporal stores keep the column tables in the L2 (pxl_reproject_dma.h).  Measured (profiles/r03_tune_nt.txt): +0.5-0.7 % on
    // the 22 GB same-resolution IQU launch and +1.5 % when down-sampling 2x, but -1.7 % at 2x refinement and -2.4 ... -5 % on
    // launches of a few GB (a 1/8 strip, the 1' map), whose tables stay in the L2 anyway: -1 = by launch size at execute time
    // Round 4 (profiles/r04_tune_rh_nt.txt, the kernel with the prefetch distance counted in source rows): at 2x refinement and beyond
    // non-temporal stores now WIN wherever the output is not cache resident -- +17 % on a 1/8 declination strip of the 1' -> 0.5'
    // refinement (0.213 -> 0.182 ms: what a rank of a sharded job runs), +2.7 % at 4x, +1.3 % on the IQU map onto the 0.25' grid,
    // +0.3 % on the whole 1' map in a two-class placement (equal within noise in a one-class one); the 268 MB output of config 2
    // measures the same either way: -2 = refinement rule at execute time (nt when the launch writes >= 512 MB)
    // ... and measured in bursts (the steady state of a caller repeating one plan, which is also what bench.py times) non-temporal
    // stores win on EVERY launch: 1/8 strip of the 2x refinement 0.225 -> 0.180 ms, config 2 (cache resident) 0.079 -> 0.064 ms, the 1'
    // same-resolution map 0.664 -> 0.656 ms, a 1/8 strip of the IQU map 1.029 -> 1.017 ms, 4x refinement +2.3 %.  The earlier "-2.4 ...
    // -5 % on launches of a few GB" was the interleaving artefact: an nt launch timed behind a plain-store launch pays for that
    // launch's dirty lines.  1 = always.
    pl->nt = env_int("PXL_REPROJECT_NT", 1);
    if (pl->pf < 0) pl->pf = 0;
    const int max_seg = PXL_MAXCH * 128;
    auto seg_for = [&](int pairs) -> int64_t {
        // footprint of TW columns: ceil(TW*sx) cells + 2 (tap +1, rounding) + 1 (even alignment) + 2 slack
        double span = ceil((double)(128 * pairs) * sx) + 5.0;
        int64_t s = (int64_t)span;
        return (s + 1) & ~(int64_t)1;
    };
    // lane width: 2 pairs (256 columns per wave) at equal resolution; up-sampling in RA takes 512 columns per wave
    // (4 KB contiguous per store row, half the column-halo re-reads): +15 % at >= ~3x (10800 -> 43200) and, since the
    // row loop keeps one interpolant per source row in registers, +2 % at 2x (round 3: 1.794 vs 1.830 ms on the
    // 1' -> 0.5' map in a slow placement, 1.631 vs 1.638 ms in a fast one; profiles/r03_tune_cfg3_a.txt)
    int want = env_int("PXL_REPROJECT_PAIRS", sx <= 0.55 ? 4 : 2);
    if (want != 1 && want != 2 && want != 4) want = 2;
    // LDS-DMA kernel: widest lane width whose slot fits
    pl->pairs_dma = want;
    while (pl->pairs_dma > 1 && seg_for(pl->pairs_dma) > max_seg) pl->pairs_dma >>= 1;
    pl->seg_dma = (int)std::min<int64_t>(seg_for(pl->pairs_dma), max_seg);
    // register-staged kernel: 1 or 2
    pl->pairs = want > 2 ? 2 : want;
    if (seg_for(pl->pairs) > max_seg && pl->pairs == 2) pl->pairs = 1;
    int64_t seg = seg_for(pl->pairs);
    // stageable: the slot fits, never laps the ring of pixels, and rows are not skipped wholesale
    auto stageable = [&](int64_t sg) { return (sg <= max_seg) && !(pl->periodic && sg > pl->nx) && sy <= 3.0; };
    pl->staged_ok = stageable(seg) && stageable(seg_for(pl->pairs_dma));
    pl->seg = (int)(seg <= max_seg ? seg : max_seg);
    {   // Float32 storage: a wave access covers 256 elements, slots hold up to 5 x 256
        const int max_seg32 = PXL_MAXCH * 256;
        auto seg_for32 = [&](int pairs) -> int64_t {
            double span = ceil((double)(256 * pairs) * sx) + 8.0;      // + taps, rounding and up to 3 of alignment
            return ((int64_t)span + 3) & ~(int64_t)3;
        };
        // 4 pixels per lane make up-sampling VALU-heavy in Float32: narrower tiles there (measured 0.98 vs 1.18 ms)
        pl->pairs_dma32 = env_int("PXL_REPROJECT_PAIRS", sx < 0.75 ? 1 : 2);
        if (pl->pairs_dma32 != 1 && pl->pairs_dma32 != 2 && pl->pairs_dma32 != 4) pl->pairs_dma32 = 2;
        while (pl->pairs_dma32 > 1 && seg_for32(pl->pairs_dma32) > max_seg32) pl->pairs_dma32 >>= 1;
        int64_t s32 = seg_for32(pl->pairs_dma32);
        pl->dma32_ok = (s32 <= max_seg32) && !(pl->periodic && s32 > pl->nx) && sy <= 3.0 && (pl->nx % 4 == 0);
        pl->seg_dma32 = (int)std::min<int64_t>(s32, max_seg32);
    }
    pl->vec_load = (pl->nx % 2 == 0);
    *out = pl;
    return PXL_OK;
}

int pxl_reproject_plan_set_variant(pxl_reproject_plan* pl, int variant) {
    if (!pl || variant < 0 || variant > 2) return fail(PXL_EINVAL, "set_variant: bad argument");
    pl->variant = variant;
    return PXL_OK;
}

int pxl_reproject_plan_destroy(pxl_reproject_plan* pl) {
    if (!pl) return PXL_OK;
    if (pl->comm_stream) {
        (void)hipStreamSynchronize(pl->comm_stream);
        (void)hipEventDestroy(pl->ev_ready);
        (void)hipEventDestroy(pl->ev_halo);
        (void)hipStreamDestroy(pl->comm_stream);
    }
    if (pl->table_mem) (void)hipFree(pl->table_mem);
    delete[] pl->h_yj0;
    delete pl;
    return PXL_OK;
}

int pxl_reproject_build_tables(pxl_reproject_plan* pl, void* stream) {
    if (!pl) return fail(PXL_EINVAL, "build_tables: null plan");
    CarAffine co = car_affine(pl->wout);
    Sky2Pix si = sky2pix_setup(pl->win, pl->nx, pl->ny, 1, PXL_FORM_DIV);
    hipLaunchKernelGGL(k_build_tables, dim3(stream_grid(pl->nxo + pl->nyo, 256)), dim3(256), 0,
                       (hipStream_t)stream, co, si, pl->nxo, pl->nyo, pl->xi0, pl->xfx, pl->yj0, pl->yfy);
    int rc = check_launch("k_build_tables");
    if (rc == PXL_OK) pl->tables_built = true;
    return rc;
}

// dtype: 8 = Float64 storage, 4 = Float32 storage (coordinates and weights are Float64 either way)
static int reproject_rows_impl(pxl_reproject_plan* pl, const void* src, void* dst, int64_t r0, int64_t nr,
                               void* stream, int dtype) {
    if (!pl) return fail(PXL_EINVAL, "execute: null plan");
    if (r0 < 0 || nr < 0 || r0 + nr > pl->dst_nrows) return fail(PXL_EINVAL, "execute: rows outside the dst window");
    if (nr == 0) return PXL_OK;
    if (!dst || (!src && pl->src_nrows > 0)) return fail(PXL_EINVAL, "execute: null src/dst");
    if (!pl->tables_built) return fail(PXL_EINVAL, "execute_rows: tables not built");
    hipStream_t st = (hipStream_t)stream;
    const bool f32 = dtype == 4;

    ReprojParams p;
    memset(&p, 0, sizeof(p));
    p.src = src; p.dst = dst;
    p.xi0 = pl->xi0; p.xfx = pl->xfx; p.yj0 = pl->yj0; p.yfy = pl->yfy;
    p.nx = pl->nx; p.ny = pl->ny; p.src_row0 = pl->src_row0; p.src_nrows = pl->src_nrows;
    p.nxo = pl->nxo; p.dst_row0 = pl->dst_row0; p.dst_nrows = pl->dst_nrows;
    p.r0 = r0; p.nr = nr; p.nc = (int32_t)pl->nc; p.periodic = pl->periodic;

    const bool aligned = (((uintptr_t)src & 15) == 0);
    bool staged = f32 ? (pl->dma32_ok && aligned) : pl->staged_ok;
    if (pl->variant == 1) staged = false;
    if (!staged) {
        int64_t work = ((pl->nxo + 1) / 2) * nr;
        dim3 g(stream_grid(work, 256), (unsigned)pl->nc);
        if (f32) hipLaunchKernelGGL((k_reproject_gather<float>), g, dim3(256), 0, st, p);
        else     hipLaunchKernelGGL((k_reproject_gather<double>), g, dim3(256), 0, st, p);
        return check_launch("k_reproject_gather");
    }

    const bool vec = f32 ? true : (pl->vec_load && aligned);
    const bool use_dma = f32 ? true : (vec && pl->variant != 2);
    const int pairs = f32 ? pl->pairs_dma32 : (use_dma ? pl->pairs_dma : pl->pairs);
    const int cw = f32 ? 256 : 128;                   // elements per wave access
    const int TW = cw * pairs;
    p.seg = f32 ? pl->seg_dma32 : (use_dma ? pl->seg_dma : pl->seg);
    p.dxpos = pl->dxpos; p.dypos = pl->dypos; p.flags = pl->flags;
    p.ntx = (int32_t)((pl->nxo + TW - 1) / TW);
    // tile height: the configured rh, halved while the launch would leave the chip short of waves (fewer than min_tiles tiles);
    // small maps and thin strips get shorter tiles
    int rh = f32 ? pl->rh32 : pl->rh;
    while (rh > 4 && (int64_t)p.ntx * ((nr + rh - 1) / rh) * pl->nc < pl->min_tiles) rh >>= 1;
    p.rh = rh;
    p.nty = (int32_t)((nr + rh - 1) / rh);
    p.ntiles = (int64_t)p.ntx * p.nty * pl->nc;
    p.tiles_per_xcd = (p.ntiles + 7) / 8;
    // PXL_REPROJECT_XCHUNK: tiles an XCD takes in one piece (0 = one contiguous eighth of the launch per XCD)
    p.xchunk = pl->xchunk > 0 && pl->xchunk < p.tiles_per_xcd ? pl->xchunk : p.tiles_per_xcd;
    int64_t nblocks = (p.ntiles + 8 * p.xchunk - 1) / (8 * p.xchunk) * (8 * p.xchunk);
    if (nblocks > 0x7fffffffLL) return fail(PXL_EINVAL, "execute: too many tiles (%lld)", (long long)nblocks);
    dim3 grid((unsigned)nblocks), block(64);
    if (use_dma) {
        // LDS-DMA fast path; shrink the ring if it would not fit a CU's LDS comfortably
        const size_t esz = f32 ? 4 : 8;
        p.ns = pl->ns; p.pf = pl->pf; p.zero_page = pl->zero_page;
        {
            const double wbytes = (double)nr * (double)pl->nxo * (double)pl->nc * (f32 ? 4.0 : 8.0);
            p.nt = pl->nt >= 0 ? pl->nt : (wbytes >= (pl->nt == -2 ? 512e6 : 12e9) ? 1 : 0);
        }
        while ((size_t)p.ns * p.seg * esz > (size_t)pl->ring_kb * 1024 && p.ns > 4) p.ns >>= 1;   // 17 KiB: >= 9 waves per CU
        size_t dma_lds = (size_t)p.ns * (size_t)p.seg * esz;
        const int nch = (p.seg + cw - 1) / cw;
        if (f32) return launch_reproject_dma_t<float>(pairs, nch, grid, dma_lds, st, p);
        return launch_reproject_dma_t<double>(pairs, nch, grid, dma_lds, st, p);
    }
    size_t lds_bytes = (size_t)PXL_NS * (size_t)pl->seg * sizeof(double);
    if (pl->pairs == 2) {
        if (vec) hipLaunchKernelGGL((k_reproject_staged<2, true>), grid, block, lds_bytes, st, p);
        else     hipLaunchKernelGGL((k_reproject_staged<2, false>), grid, block, lds_bytes, st, p);
    } else {
        if (vec) hipLaunchKernelGGL((k_reproject_staged<1, true>), grid, block, lds_bytes, st, p);
        else     hipLaunchKernelGGL((k_reproject_staged<1, false>), grid, block, lds_bytes, st, p);
    }
    return check_launch("k_reproject_staged");
}

int pxl_reproject_execute_rows(pxl_reproject_plan* pl, const double* src, double* dst, int64_t r0, int64_t nr,
                               void* stream) {
    return reproject_rows_impl(pl, src, dst, r0, nr, stream, 8);
}

int pxl_reproject_execute_rows_f32(pxl_reproject_plan* pl, const float* src, float* dst, int64_t r0, int64_t nr,
                                   void* stream) {
    return reproject_rows_impl(pl, src, dst, r0, nr, stream, 4);
}

int pxl_reproject_execute(pxl_reproject_plan* pl, const double* src, double* dst, void* stream) {
    int rc = pxl_reproject_build_tables(pl, stream);
    if (rc) return rc;
    return pxl_reproject_execute_rows(pl, src, dst, 0, pl->dst_nrows, stream);
}

int pxl_reproject_execute_f32(pxl_reproject_plan* pl, const float* src, float* dst, void* stream) {
    int rc = pxl_reproject_build_tables(pl, stream);
    if (rc) return rc;
    return pxl_reproject_execute_rows_f32(pl, src, dst, 0, pl->dst_nrows, stream);
}

int pxl_reproject_plan_src_rows(const pxl_reproject_plan* pl, int64_t* lo, int64_t* hi) {
    if (!pl || !lo || !hi) return fail(PXL_EINVAL, "plan_src_rows: null argument");
    int64_t l = INT64_MAX, h = INT64_MIN;
    for (int64_t r = 0; r < pl->dst_nrows; ++r) {
        int64_t j0 = pl->h_yj0[pl->dst_row0 + r];
        for (int64_t j = j0; j <= j0 + 1; ++j)
            if (j >= 1 && j <= pl->ny) { if (j - 1 < l) l = j - 1; if (j > h) h = j; }
    }
    if (l > h) { l = 0; h = 0; }
    *lo = l; *hi = h;
    return PXL_OK;
}

int pxl_reproject_plan_rows_covered(const pxl_reproject_plan* pl, int64_t have_lo, int64_t have_hi, int64_t* lo,
                                    int64_t* hi) {
    if (!pl || !lo || !hi) return fail(PXL_EINVAL, "plan_rows_covered: null argument");
    // longest run of output rows (relative) whose in-map taps j0, j0+1 all lie in [have_lo, have_hi)
    int64_t best_lo = 0, best_hi = 0, cur_lo = -1;
    for (int64_t r = 0; r <= pl->dst_nrows; ++r) {
        bool ok = false;
        if (r < pl->dst_nrows) {
            int64_t j0 = pl->h_yj0[pl->dst_row0 + r];
            ok = true;
            for (int64_t j = j0; j <= j0 + 1; ++j)
                if (j >= 1 && j <= pl->ny && !(j - 1 >= have_lo && j - 1 < have_hi)) ok = false;
        }
        if (ok) { if (cur_lo < 0) cur_lo = r; }
        else if (cur_lo >= 0) {
            if (r - cur_lo > best_hi - best_lo) { best_lo = cur_lo; best_hi = r; }
            cur_lo = -1;
        }
    }
    *lo = best_lo; *hi = best_hi;
    return PXL_OK;
}

// ---- sharded step: halo rows over RCCL send/recv, interior rows while they travel, boundary rows after
static int sharded_step_impl(pxl_reproject_plan* pl, void* src, void* dst, int64_t own_row0, int64_t own_nrows,
                             const pxl_halo_xfer* sends, int nsends, const pxl_halo_xfer* recvs, int nrecvs,
                             void* comm, void* stream, int dtype) {
    if (!pl || !src || !dst) return fail(PXL_EINVAL, "sharded_step: null plan or buffer");
    if (nsends < 0 || nrecvs < 0 || (nsends > 0 && !sends) || (nrecvs > 0 && !recvs)) return fail(PXL_EINVAL, "sharded_step: bad transfer lists");
    const int64_t lo = pl->src_row0, hi = pl->src_row0 + pl->src_nrows;
    if (own_row0 < lo || own_nrows < 0 || own_row0 + own_nrows > hi) return fail(PXL_EINVAL, "sharded_step: owned rows outside the plan's source window");
    hipStream_t st = (hipStream_t)stream;
    int rc = PXL_OK;
    const bool exchange = nsends + nrecvs > 0;
    if (exchange) {
        if (!comm) return fail(PXL_EINVAL, "sharded_step: transfers listed but no RCCL communicator");
        const RcclApi& nc = rccl_api(false);
        if (!nc.ok) return fail(PXL_ENODEV, "sharded_step: RCCL entry points not available: %s", nc.where);
        if (nc.own && !rccl_own_comm_has(comm))
            return fail(PXL_ENODEV, "sharded_step: the communicator was not created by pxl_comm_init_rank, and the only RCCL instance "
                                    "this library can see is one it loaded itself (%s): a foreign communicator belongs to another instance", nc.where);
        int nranks = 0, me = -1;
        if (nc.CommCount((ncclComm_t)comm, &nranks) != ncclSuccess || nc.CommUserRank((ncclComm_t)comm, &me) != ncclSuccess)
            return fail(PXL_EINVAL, "sharded_step: not a usable RCCL communicator");
        for (int pass = 0; pass < 2; ++pass) {
            const pxl_halo_xfer* x = pass == 0 ? sends : recvs;
            for (int i = 0; i < (pass == 0 ? nsends : nrecvs); ++i) {
                if (x[i].peer < 0 || x[i].peer >= nranks) return fail(PXL_EINVAL, "sharded_step: peer %d outside the communicator (%d ranks)", x[i].peer, nranks);
                if (x[i].nrows < 1 || x[i].row0 < lo || x[i].row0 + x[i].nrows > hi) return fail(PXL_EINVAL, "sharded_step: transfer rows outside the plan's source window");
                if (pass == 0 && (x[i].row0 < own_row0 || x[i].row0 + x[i].nrows > own_row0 + own_nrows))
                    return fail(PXL_EINVAL, "sharded_step: a rank can only send rows it owns");
            }
        }
        if (!pl->comm_stream) {
            HIP_TRY(hipStreamCreateWithFlags(&pl->comm_stream, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&pl->ev_ready, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&pl->ev_halo, hipEventDisableTiming));
        }
        // the exchange may start once everything queued on the caller's stream so far (the producers of src, the
        // previous step's readers of the halo rows) is done
        HIP_TRY(hipEventRecord(pl->ev_ready, st));
        HIP_TRY(hipStreamWaitEvent(pl->comm_stream, pl->ev_ready, 0));
        const size_t esz = (size_t)dtype;
        const ncclDataType_t dt = dtype == 4 ? ncclFloat32 : ncclFloat64;
        ncclResult_t r = nc.GroupStart();
        // one message per component plane and transfer: rows of one plane are contiguous in the resident buffer,
        // so nothing is staged
        for (int i = 0; i < nsends && r == ncclSuccess; ++i)
            for (int64_t c = 0; c < pl->nc && r == ncclSuccess; ++c)
                r = nc.Send((const char*)src + ((c * pl->src_nrows + (sends[i].row0 - lo)) * pl->nx) * esz,
                            (size_t)(sends[i].nrows * pl->nx), dt, sends[i].peer, (ncclComm_t)comm, pl->comm_stream);
        for (int i = 0; i < nrecvs && r == ncclSuccess; ++i)
            for (int64_t c = 0; c < pl->nc && r == ncclSuccess; ++c)
                r = nc.Recv((char*)src + ((c * pl->src_nrows + (recvs[i].row0 - lo)) * pl->nx) * esz,
                            (size_t)(recvs[i].nrows * pl->nx), dt, recvs[i].peer, (ncclComm_t)comm, pl->comm_stream);
        const ncclResult_t rend = nc.GroupEnd();
        if (r == ncclSuccess) r = rend;
        if (r != ncclSuccess) return fail(PXL_EHIP, "sharded_step: RCCL send/recv failed: %s", nc.GetErrorString(r));
        HIP_TRY(hipEventRecord(pl->ev_halo, pl->comm_stream));
    }
    rc = pxl_reproject_build_tables(pl, stream);
    if (rc) return rc;
    // rows computable from the rows this rank owns run while the halo is in flight
    int64_t i_lo = 0, i_hi = pl->dst_nrows;
    if (exchange) {
        if (pl->cov_have_lo != own_row0 || pl->cov_have_hi != own_row0 + own_nrows || pl->cov_hi < pl->cov_lo) {
            rc = pxl_reproject_plan_rows_covered(pl, own_row0, own_row0 + own_nrows, &pl->cov_lo, &pl->cov_hi);
            if (rc) return rc;
            pl->cov_have_lo = own_row0; pl->cov_have_hi = own_row0 + own_nrows;
        }
        i_lo = pl->cov_lo; i_hi = pl->cov_hi;
    }
    if (i_hi > i_lo) {
        rc = reproject_rows_impl(pl, src, dst, i_lo, i_hi - i_lo, stream, dtype);
        if (rc) return rc;
    }
    if (exchange) {
        HIP_TRY(hipStreamWaitEvent(st, pl->ev_halo, 0));
        if (i_hi > i_lo) {
            if (i_lo > 0) { rc = reproject_rows_impl(pl, src, dst, 0, i_lo, stream, dtype); if (rc) return rc; }
            if (i_hi < pl->dst_nrows) rc = reproject_rows_impl(pl, src, dst, i_hi, pl->dst_nrows - i_hi, stream, dtype);
        } else {
            rc = reproject_rows_impl(pl, src, dst, 0, pl->dst_nrows, stream, dtype);
        }
    }
    return rc;
}

// ---- communicator helpers for hosts without an RCCL binding of their own (Julia, C): thin wrappers over
// ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy, resolved through pxl_rccl.h
int pxl_comm_unique_id(void* id128) {
    if (!id128) return fail(PXL_EINVAL, "comm_unique_id: null buffer (PXL_COMM_ID_BYTES bytes)");
    const RcclApi& nc = rccl_api(true);
    if (!nc.ok) return fail(PXL_ENODEV, "comm_unique_id: RCCL not available: %s", nc.where);
    static_assert(sizeof(ncclUniqueId) == PXL_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    ncclResult_t r = nc.GetUniqueId(&id);
    if (r != ncclSuccess) return fail(PXL_EHIP, "comm_unique_id: ncclGetUniqueId: %s", nc.GetErrorString(r));
    memcpy(id128, &id, sizeof id);
    return PXL_OK;
}

int pxl_comm_init_rank(const void* id128, int rank, int nranks, void** comm) {
    if (!id128 || !comm) return fail(PXL_EINVAL, "comm_init_rank: null argument");
    *comm = nullptr;
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(PXL_EINVAL, "comm_init_rank: rank %d outside [0, %d)", rank, nranks);
    const RcclApi& nc = rccl_api(true);
    if (!nc.ok) return fail(PXL_ENODEV, "comm_init_rank: RCCL not available: %s", nc.where);
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclComm_t c = nullptr;
    ncclResult_t r = nc.CommInitRank(&c, nranks, id, rank);          // collective over the nranks callers; uses the current device
    if (r != ncclSuccess) return fail(PXL_EHIP, "comm_init_rank: ncclCommInitRank: %s", nc.GetErrorString(r));
    rccl_own_comm_add((void*)c);
    *comm = (void*)c;
    return PXL_OK;
}

int pxl_comm_destroy(void* comm) {
    if (!comm) return PXL_OK;
    if (!rccl_own_comm_erase(comm)) return fail(PXL_EINVAL, "comm_destroy: not a communicator created by pxl_comm_init_rank");
    const RcclApi& nc = rccl_api(false);
    if (!nc.ok) return fail(PXL_ENODEV, "comm_destroy: RCCL not available: %s", nc.where);
    ncclResult_t r = nc.CommDestroy((ncclComm_t)comm);
    if (r != ncclSuccess) return fail(PXL_EHIP, "comm_destroy: ncclCommDestroy: %s", nc.GetErrorString(r));
    return PXL_OK;
}

const char* pxl_comm_backend(void) {
    // a diagnostic: it probes for an instance already in the process but latches nothing (pxl_rccl.h); the text lives in
    // a thread-local buffer, valid until this thread asks again
    static thread_local char where[256];
    const RcclApi nc = rccl_api(false);
    snprintf(where, sizeof where, "%s", nc.where);
    return where;
}

int pxl_reproject_sharded_step_f64(pxl_reproject_plan* plan, double* src, double* dst, int64_t own_row0, int64_t own_nrows,
                                   const pxl_halo_xfer* sends, int nsends, const pxl_halo_xfer* recvs, int nrecvs,
                                   void* rccl_comm, void* stream) {
    return sharded_step_impl(plan, src, dst, own_row0, own_nrows, sends, nsends, recvs, nrecvs, rccl_comm, stream, 8);
}

int pxl_reproject_sharded_step_f32(pxl_reproject_plan* plan, float* src, float* dst, int64_t own_row0, int64_t own_nrows,
                                   const pxl_halo_xfer* sends, int nsends, const pxl_halo_xfer* recvs, int nrecvs,
                                   void* rccl_comm, void* stream) {
    return sharded_step_impl(plan, src, dst, own_row0, own_nrows, sends, nsends, recvs, nrecvs, rccl_comm, stream, 4);
}

int pxl_reproject_car_bilinear_f64(const pxl_car_wcs* wcs_in, const int64_t shape_in[3], const double* src,
                                   const pxl_car_wcs* wcs_out, const int64_t shape_out[2], double* dst,
                                   void* stream) {
    if (!shape_in || !shape_out) return fail(PXL_EINVAL, "reproject: null shape");
    pxl_reproject_plan* pl = nullptr;
    int rc = pxl_reproject_plan_create(wcs_in, shape_in, 0, shape_in[1], wcs_out, shape_out, 0, shape_out[1], &pl);
    if (rc) return rc;
    rc = pxl_reproject_execute(pl, src, dst, stream);
    if (rc == PXL_OK) {
        hipError_t e = hipStreamSynchronize((hipStream_t)stream);
        if (e != hipSuccess) rc = fail(PXL_EHIP, "reproject: %s", hipGetErrorString(e));
    }
    pxl_reproject_plan_destroy(pl);
    return rc;
}

// diagnostics of the tiled generic reprojection: how many 128 x 32 tiles of the last call took the exact path
// Two slots used alternately: a call counts in one, and its last launch zeroes the other for the next call (no memset per call).
static unsigned int* g_exact_tiles[64] = {};
static int g_exact_slot[64] = {};
static int64_t g_generic_tiles[64] = {};
static unsigned int* exact_tiles_counter(int* dev_out, bool advance = false) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    *dev_out = dev;
    std::lock_guard<std::mutex> lock(g_pool_mu);
    if (!g_exact_tiles[dev]) {
        if (hipMalloc((void**)&g_exact_tiles[dev], 64) != hipSuccess || hipMemset(g_exact_tiles[dev], 0, 64) != hipSuccess) {
            (void)hipGetLastError(); g_exact_tiles[dev] = nullptr; return nullptr;
        }
    }
    if (advance) g_exact_slot[dev] ^= 1;
    return g_exact_tiles[dev] + g_exact_slot[dev];
}

int pxl_reproject_generic_last_tiles(int64_t* exact_tiles, int64_t* total_tiles, void* stream) {
    if (!exact_tiles || !total_tiles) return fail(PXL_EINVAL, "generic_last_tiles: null argument");
    int dev = 0;
    unsigned int* c = exact_tiles_counter(&dev);
    if (!c) return fail(PXL_ENODEV, "generic_last_tiles: no counter on this device");
    unsigned int v = 0;
    HIP_TRY(hipMemcpyAsync(&v, c, 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    *exact_tiles = v; *total_tiles = g_generic_tiles[dev];
    return PXL_OK;
}

// geometry part of GenericParams (everything but the buffers and the component count)
static int generic_params(const char* who, const pxl_car_wcs* wcs_in, int proj_in, const int64_t* shape_in,
                          const pxl_car_wcs* wcs_out, int proj_out, const int64_t* shape_out, GenericParams* out) {
    if (!wcs_ok(wcs_in) || !wcs_ok(wcs_out)) return fail(PXL_EINVAL, "%s: invalid WCS", who);
    if (!shape_in || !shape_out) return fail(PXL_EINVAL, "%s: null shape", who);
    if (shape_in[0] < 1 || shape_in[1] < 1 || shape_out[0] < 1 || shape_out[1] < 1)
        return fail(PXL_EINVAL, "%s: shapes must be positive", who);
    if ((proj_in != PXL_PROJ_CAR && proj_in != PXL_PROJ_TAN) || (proj_out != PXL_PROJ_CAR && proj_out != PXL_PROJ_TAN))
        return fail(PXL_EINVAL, "%s: unknown projection code", who);
    GenericParams p;
    memset(&p, 0, sizeof(p));
    p.nx = shape_in[0]; p.ny = shape_in[1]; p.nc = 1;
    p.nxo = shape_out[0]; p.nyo = shape_out[1];
    p.proj_in = proj_in; p.proj_out = proj_out;
    p.periodic = (proj_in == PXL_PROJ_CAR) &&
                 fabs((double)p.nx * fabs(wcs_in->cdelt[0] * wcs_in->unit) - PXL_TWOPI_D) < 1e-8;
    if (proj_out == PXL_PROJ_TAN) p.out_tan = tan_setup(*wcs_out); else p.out_car = car_affine(*wcs_out);
    if (proj_in == PXL_PROJ_TAN) p.in_tan = tan_setup(*wcs_in);
    else p.in_car = sky2pix_setup(*wcs_in, p.nx, p.ny, 1, PXL_FORM_DIV);
    *out = p;
    return PXL_OK;
}
static void launch_generic_pixels(const GenericParams& p, int64_t gx, int64_t gy, const double2* lat, const int32_t* flag, hipStream_t st) {
    // PXL_GENERIC_V=1: round 3's pixel kernel (one pixel per lane, 8-byte taps, ~99 VALU per pixel), kept for A/B; default: the lean
    // round-4 form.  (Two more forms were built, measured slower and moved to tools/research/: two pixels per lane, the LDS ring.)
    if (env_int("PXL_GENERIC_V", 3) == 1 || env_int("PXL_GENERIC_V1", 0))
        hipLaunchKernelGGL(k_reproject_generic_tiled, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, st, p, lat, flag);
    else
        hipLaunchKernelGGL(k_reproject_generic_tiled3, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, st, p, lat, flag);
}

// ---- the generic operator with its lattice kept (include/pixell_hip.h)
struct pxl_generic_plan {
    GenericParams p;             // geometry; src / dst / nc are filled per execute
    int64_t gx, gy, ntiles, exact;
    char* ws;                    // lattice (ntiles x 42 coordinate pairs) + flags (ntiles x int32) + the counter of flagged tiles
    double2* lat; int32_t* flag; unsigned int* counter;
    int device;
};

int pxl_generic_plan_create(const pxl_car_wcs* wcs_in, int proj_in, const int64_t shape_in[2],
                            const pxl_car_wcs* wcs_out, int proj_out, const int64_t shape_out[2],
                            void* stream, pxl_generic_plan** plan) {
    if (!plan) return fail(PXL_EINVAL, "generic_plan_create: null result");
    *plan = nullptr;
    GenericParams p;
    int rc = generic_params("generic_plan_create", wcs_in, proj_in, shape_in, wcs_out, proj_out, shape_out, &p);
    if (rc) return rc;
    const int64_t gx = (p.nxo + PXL_TW - 1) / PXL_TW, gy = (p.nyo + PXL_TH - 1) / PXL_TH;
    if (gy > 65535) return fail(PXL_EINVAL, "generic_plan_create: more than 65 535 tile rows (use the one-shot entry)");
    const int64_t ntiles = gx * gy;
    const size_t lat_bytes = (size_t)ntiles * (PXL_TNX * PXL_TNY) * sizeof(double2), flag_bytes = ((size_t)ntiles * 4 + 15) & ~(size_t)15;
    pxl_generic_plan* pl = new (std::nothrow) pxl_generic_plan();
    if (!pl) return fail(PXL_ENOMEM, "generic_plan_create: out of host memory");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipGetDevice(&pl->device);
    if (e == hipSuccess) e = hipMalloc((void**)&pl->ws, lat_bytes + flag_bytes + 16);
    if (e != hipSuccess) { (void)hipGetLastError(); delete pl; return fail(PXL_EHIP, "generic_plan_create: %s", hipGetErrorString(e)); }
    pl->lat = (double2*)pl->ws; pl->flag = (int32_t*)(pl->ws + lat_bytes); pl->counter = (unsigned int*)(pl->ws + lat_bytes + flag_bytes);
    pl->gx = gx; pl->gy = gy; pl->ntiles = ntiles;
    p.exact_tiles = pl->counter; p.exact_tiles_next = nullptr;
    pl->p = p;
    unsigned int host_count = 0;
    e = hipMemsetAsync(pl->counter, 0, 16, st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_generic_lattice, dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0, st, p, gx, ntiles, pl->lat, pl->flag);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&host_count, pl->counter, 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { (void)hipFree(pl->ws); delete pl; return fail(PXL_EHIP, "generic_plan_create: %s", hipGetErrorString(e)); }
    pl->exact = host_count;
    *plan = pl;
    return PXL_OK;
}

int pxl_generic_plan_execute(const pxl_generic_plan* plan, int64_t ncomp, const double* src, double* dst, void* stream) {
    if (!plan) return fail(PXL_EINVAL, "generic_plan_execute: null plan");
    if (ncomp < 1 || ncomp > 0x7fffffff) return fail(PXL_EINVAL, "generic_plan_execute: component count must be positive");
    if (!src || !dst) return fail(PXL_EINVAL, "generic_plan_execute: null src/dst");
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != plan->device) return fail(PXL_EINVAL, "generic_plan_execute: plan belongs to device %d, current device is %d", plan->device, dev);
    GenericParams p = plan->p;
    p.src = src; p.dst = dst; p.nc = (int32_t)ncomp;
    hipStream_t st = (hipStream_t)stream;
    launch_generic_pixels(p, plan->gx, plan->gy, plan->lat, plan->flag, st);
    if (plan->exact > 0)       // known on the host since the plan was made: no launch at all for the usual patch
        hipLaunchKernelGGL(k_reproject_generic_exact_tiles, dim3((unsigned)std::min<int64_t>(plan->ntiles, 256)), dim3(256), 0, st, p, (const int32_t*)plan->flag, plan->gx, plan->ntiles);
    return check_launch("k_reproject_generic_tiled (plan)");
}

int pxl_generic_plan_tiles(const pxl_generic_plan* plan, int64_t* exact_tiles, int64_t* total_tiles) {
    if (!plan || !exact_tiles || !total_tiles) return fail(PXL_EINVAL, "generic_plan_tiles: null argument");
    *exact_tiles = plan->exact; *total_tiles = plan->ntiles;
    return PXL_OK;
}

int pxl_generic_plan_destroy(pxl_generic_plan* plan) {
    if (!plan) return PXL_OK;
    hipError_t e = plan->ws ? hipFree(plan->ws) : hipSuccess;
    delete plan;
    if (e != hipSuccess) return fail(PXL_EHIP, "generic_plan_destroy: %s", hipGetErrorString(e));
    return PXL_OK;
}

int pxl_reproject_generic_bilinear_f64(const pxl_car_wcs* wcs_in, int proj_in, const int64_t shape_in[3],
                                       const double* src, const pxl_car_wcs* wcs_out, int proj_out,
                                       const int64_t shape_out[2], double* dst, void* stream) {
    GenericParams p;
    int rcp = generic_params("reproject_generic", wcs_in, proj_in, shape_in, wcs_out, proj_out, shape_out, &p);
    if (rcp) return rcp;
    if (shape_in[2] < 1) return fail(PXL_EINVAL, "reproject_generic: shapes must be positive");
    if (!src || !dst) return fail(PXL_EINVAL, "reproject_generic: null src/dst");
    p.src = src; p.dst = dst; p.nc = (int32_t)shape_in[2];
    // PXL_GENERIC_EXACT=1: per-pixel evaluation of the coordinates (the definition; cross-check and fallback of the
    // tiled kernel, which interpolates them per 128 x 32 tile within PXL_TILED_TOL pixel)
    const int64_t gx = (p.nxo + PXL_TW - 1) / PXL_TW, gy = (p.nyo + PXL_TH - 1) / PXL_TH;
    if (env_int("PXL_GENERIC_EXACT", 0) || gy > 65535) {
        hipLaunchKernelGGL(k_reproject_generic, dim3(stream_grid(p.nxo * p.nyo, 256)), dim3(256), 0, (hipStream_t)stream, p);
        return check_launch("k_reproject_generic");
    }
    int dev = 0;
    p.exact_tiles = exact_tiles_counter(&dev, true);
    if (p.exact_tiles) { p.exact_tiles_next = g_exact_tiles[dev] + (g_exact_slot[dev] ^ 1); g_generic_tiles[dev] = gx * gy; }
    // per-tile lattice (30 coordinate pairs) + flag from the library's stream-ordered scratch pool
    const int64_t ntiles = gx * gy;
    const size_t lat_bytes = (size_t)ntiles * (PXL_TNX * PXL_TNY) * sizeof(double2), total_bytes = lat_bytes + (size_t)ntiles * 4;
    hipStream_t st = (hipStream_t)stream;
    char* ws = nullptr;
    hipMemPool_t pool = unwind_pool();
    if (pool) HIP_TRY(hipMallocFromPoolAsync((void**)&ws, total_bytes, pool, st));
    else HIP_TRY(hipMallocAsync((void**)&ws, total_bytes, st));
    double2* lat = (double2*)ws;
    int32_t* flag = (int32_t*)(ws + lat_bytes);
    hipLaunchKernelGGL(k_generic_lattice, dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0, st, p, gx, ntiles, lat, flag);
    launch_generic_pixels(p, gx, gy, (const double2*)lat, (const int32_t*)flag, st);
    hipLaunchKernelGGL(k_reproject_generic_exact_tiles, dim3((unsigned)std::min<int64_t>(ntiles, 256)), dim3(256), 0, st, p, (const int32_t*)flag, gx, ntiles);
    int rc = check_launch("k_reproject_generic_tiled");
    hipError_t fe = hipFreeAsync(ws, st);
    if (fe != hipSuccess && rc == PXL_OK) rc = fail(PXL_EHIP, "reproject_generic: hipFreeAsync: %s", hipGetErrorString(fe));
    return rc;
}

static int sample_impl(const pxl_car_wcs* wcs_in, const int64_t shape_in[3], const void* src, int64_t src_row0,
                       int64_t src_nrows, int64_t n, const double* sky, void* out, void* stream, int dtype) {
    if (!wcs_ok(wcs_in) || !shape_in) return fail(PXL_EINVAL, "sample: invalid WCS/shape");
    if (shape_in[0] < 1 || shape_in[1] < 1 || shape_in[2] < 1) return fail(PXL_EINVAL, "sample: shapes must be positive");
    if (src_row0 < 0 || src_nrows < 0 || src_row0 + src_nrows > shape_in[1])
        return fail(PXL_EINVAL, "sample: source window outside the map");
    if (n < 0 || (n > 0 && (!sky || !out || (!src && src_nrows > 0)))) return fail(PXL_EINVAL, "sample: null buffer or negative n");
    if (((uintptr_t)sky & 15) != 0) return fail(PXL_EINVAL, "sample: 2xN buffer must be 16-byte aligned");
    if (n == 0) return PXL_OK;
    Sky2Pix s = sky2pix_setup(*wcs_in, shape_in[0], shape_in[1], 1, PXL_FORM_RECIP);
    int periodic = fabs((double)shape_in[0] * fabs(wcs_in->cdelt[0] * wcs_in->unit) - PXL_TWOPI_D) < 1e-8;
    dim3 grid(stream_grid((n + PXL_SUNR - 1) / PXL_SUNR, 256));
    if (dtype == 4)
        hipLaunchKernelGGL((k_sample_bilinear<float>), grid, dim3(256), 0, (hipStream_t)stream, s, (const float*)src,
                           shape_in[0], shape_in[1], (int32_t)shape_in[2], src_row0, src_nrows, periodic, n,
                           (const double2*)sky, (float*)out);
    else
        hipLaunchKernelGGL((k_sample_bilinear<double>), grid, dim3(256), 0, (hipStream_t)stream, s, (const double*)src,
                           shape_in[0], shape_in[1], (int32_t)shape_in[2], src_row0, src_nrows, periodic, n,
                           (const double2*)sky, (double*)out);
    return check_launch("k_sample_bilinear");
}

int pxl_sample_car_bilinear_f64(const pxl_car_wcs* wcs_in, const int64_t shape_in[3], const double* src,
                                int64_t src_row0, int64_t src_nrows, int64_t n, const double* sky, double* out,
                                void* stream) {
    return sample_impl(wcs_in, shape_in, src, src_row0, src_nrows, n, sky, out, stream, 8);
}

int pxl_sample_car_bilinear_f32(const pxl_car_wcs* wcs_in, const int64_t shape_in[3], const float* src,
                                int64_t src_row0, int64_t src_nrows, int64_t n, const double* sky, float* out,
                                void* stream) {
    return sample_impl(wcs_in, shape_in, src, src_row0, src_nrows, n, sky, out, stream, 4);
}

// ---- row-pair layout (pxl_sample.h): caller-owned buffer of pxl_sample_pairs_elems() map elements
int64_t pxl_sample_pairs_elems(const int64_t shape_in[3], int64_t src_nrows) {
    if (!shape_in || shape_in[0] < 1 || shape_in[1] < 1 || shape_in[2] < 1 || src_nrows < 0 || src_nrows > shape_in[1]) {
        fail(PXL_EINVAL, "sample_pairs_elems: invalid shape or window");
        return -1;
    }
    // one size for both element types: Float64 rows (groups of 4 entries, 3 new columns each) are the longer ones except
    // on maps of a few columns, where the rounding up to whole groups can favour Float32 (8 entries, 7 new columns)
    const int64_t e64 = PairGroup<double>::groups(shape_in[0]) * PairGroup<double>::E;
    const int64_t e32 = PairGroup<float>::groups(shape_in[0]) * PairGroup<float>::E;
    return 2 * (e64 > e32 ? e64 : e32) * (src_nrows + 1) * shape_in[2];
}

static int build_pairs_impl(const int64_t shape_in[3], const void* src, int64_t src_nrows, void* pairs, void* stream, int dtype) {
    if (pxl_sample_pairs_elems(shape_in, src_nrows) < 0) return PXL_EINVAL;
    if (!pairs || (!src && src_nrows > 0)) return fail(PXL_EINVAL, "sample_build_pairs: null buffer");
    if (((uintptr_t)pairs & 63) != 0) return fail(PXL_EINVAL, "sample_build_pairs: pair buffer must be 64-byte aligned");
    if (shape_in[0] > 0x7fffffffLL) return fail(PXL_EINVAL, "sample_build_pairs: more than 2^31 columns");
    if (shape_in[2] > 65535) return fail(PXL_EINVAL, "sample_build_pairs: more than 65535 components");
    const int64_t nx = shape_in[0], tiles = (src_nrows + 1 + PXL_POS_ROWS - 1) / PXL_POS_ROWS;
    if (tiles > 65535) return fail(PXL_EINVAL, "sample_build_pairs: more than %lld rows per call", 65535LL * PXL_POS_ROWS);
    const int64_t pitch = dtype == 4 ? PairGroup<float>::groups(nx) * PairGroup<float>::E : PairGroup<double>::groups(nx) * PairGroup<double>::E;
    int fronts = env_int("PXL_PAIRS_FRONTS", 8);
    if (fronts < 1 || tiles < 16 * fronts) fronts = 1;
    const int64_t per = (tiles + fronts - 1) / fronts;
    if (per * fronts > 65535) fronts = 1;
    dim3 grid((unsigned)((pitch + 255) / 256), (unsigned)(fronts > 1 ? per * fronts : tiles), (unsigned)shape_in[2]);
    if (dtype == 4)
        hipLaunchKernelGGL((k_build_rowpairs<float>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)src, nx, src_nrows, (float2*)pairs, fronts);
    else
        hipLaunchKernelGGL((k_build_rowpairs<double>), grid, dim3(256), 0, (hipStream_t)stream, (const double*)src, nx, src_nrows, (double2*)pairs, fronts);
    return check_launch("k_build_rowpairs");
}

int pxl_sample_build_pairs_f64(const int64_t shape_in[3], const double* src, int64_t src_nrows, double* pairs, void* stream) {
    return build_pairs_impl(shape_in, src, src_nrows, pairs, stream, 8);
}

int pxl_sample_build_pairs_f32(const int64_t shape_in[3], const float* src, int64_t src_nrows, float* pairs, void* stream) {
    return build_pairs_impl(shape_in, src, src_nrows, pairs, stream, 4);
}

static int sample_pairs_impl(const pxl_car_wcs* wcs_in, const int64_t shape_in[3], const void* pairs, int64_t src_row0,
                             int64_t src_nrows, int64_t n, const double* sky, void* out, void* stream, int dtype) {
    if (!wcs_ok(wcs_in) || !shape_in) return fail(PXL_EINVAL, "sample_pairs: invalid WCS/shape");
    if (shape_in[0] < 1 || shape_in[1] < 1 || shape_in[2] < 1) return fail(PXL_EINVAL, "sample_pairs: shapes must be positive");
    if (src_row0 < 0 || src_nrows < 0 || src_row0 + src_nrows > shape_in[1])
        return fail(PXL_EINVAL, "sample_pairs: source window outside the map");
    if (n < 0 || (n > 0 && (!sky || !out || !pairs))) return fail(PXL_EINVAL, "sample_pairs: null buffer or negative n");
    if (((uintptr_t)sky & 15) != 0) return fail(PXL_EINVAL, "sample_pairs: 2xN buffer must be 16-byte aligned");
    if (((uintptr_t)pairs & 63) != 0) return fail(PXL_EINVAL, "sample_pairs: pair buffer must be 64-byte aligned");
    if (shape_in[0] > 0x7fffffffLL) return fail(PXL_EINVAL, "sample_pairs: more than 2^31 columns");
    if (n == 0) return PXL_OK;
    Sky2Pix s = sky2pix_setup(*wcs_in, shape_in[0], shape_in[1], 1, PXL_FORM_RECIP);
    int periodic = fabs((double)shape_in[0] * fabs(wcs_in->cdelt[0] * wcs_in->unit) - PXL_TWOPI_D) < 1e-8;
    const int unr = dtype == 4 ? PairsUnroll<float>::value : PairsUnroll<double>::value;
    dim3 grid(stream_grid((n + unr - 1) / unr, 256));
    if (dtype == 4)
        hipLaunchKernelGGL((k_sample_pairs<float>), grid, dim3(256), 0, (hipStream_t)stream, s, (const float2*)pairs,
                           shape_in[0], shape_in[1], (int32_t)shape_in[2], src_row0, src_nrows, periodic, n,
                           (const double2*)sky, (float*)out);
    else
        hipLaunchKernelGGL((k_sample_pairs<double>), grid, dim3(256), 0, (hipStream_t)stream, s, (const double2*)pairs,
                           shape_in[0], shape_in[1], (int32_t)shape_in[2], src_row0, src_nrows, periodic, n,
                           (const double2*)sky, (double*)out);
    return check_launch("k_sample_pairs");
}

int pxl_sample_car_bilinear_pairs_f64(const pxl_car_wcs* wcs_in, const int64_t shape_in[3], const double* pairs,
                                      int64_t src_row0, int64_t src_nrows, int64_t n, const double* sky, double* out,
                                      void* stream) {
    return sample_pairs_impl(wcs_in, shape_in, pairs, src_row0, src_nrows, n, sky, out, stream, 8);
}

int pxl_sample_car_bilinear_pairs_f32(const pxl_car_wcs* wcs_in, const int64_t shape_in[3], const float* pairs,
                                      int64_t src_row0, int64_t src_nrows, int64_t n, const double* sky, float* out,
                                      void* stream) {
    return sample_pairs_impl(wcs_in, shape_in, pairs, src_row0, src_nrows, n, sky, out, stream, 4);
}

int pxl_fits_decode_f64(const void* raw_be, double* dst, int64_t n, int bitpix, void* stream) {
    if (n < 0 || (n > 0 && (!raw_be || !dst))) return fail(PXL_EINVAL, "fits_decode: null buffer or negative n");
    if (bitpix != -64 && bitpix != -32) return fail(PXL_EINVAL, "fits_decode: BITPIX %d not supported (only -64, -32)", bitpix);
    if (n == 0) return PXL_OK;
    hipLaunchKernelGGL(k_bswap_to_f64, dim3(stream_grid((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, raw_be, dst, n, bitpix);
    return check_launch("k_bswap_to_f64");
}

int pxl_fits_swap_f32(const void* src, void* dst, int64_t n, void* stream) {
    if (n < 0 || (n > 0 && (!src || !dst))) return fail(PXL_EINVAL, "fits_swap_f32: null buffer or negative n");
    if (n == 0) return PXL_OK;
    hipLaunchKernelGGL(k_bswap32, dim3(stream_grid((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint32_t*)src, (uint32_t*)dst, n);
    return check_launch("k_bswap32");
}

int pxl_fits_encode_f64(const double* src, void* raw_be, int64_t n, void* stream) {
    if (n < 0 || (n > 0 && (!raw_be || !src))) return fail(PXL_EINVAL, "fits_encode: null buffer or negative n");
    if (n == 0) return PXL_OK;
    hipLaunchKernelGGL(k_f64_to_be, dim3(stream_grid((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, src, (uint64_t*)raw_be, n);
    return check_launch("k_f64_to_be");
}

int pxl_fill_random_f64(double* dst, int64_t n, uint64_t seed, uint64_t offset, int kind, void* stream) {
    if (n < 0 || (n > 0 && !dst)) return fail(PXL_EINVAL, "fill_random: null buffer or negative n");
    if (n == 0) return PXL_OK;
    hipLaunchKernelGGL(k_fill_random, dim3(stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, dst, n, seed,
                       offset, kind);
    return check_launch("k_fill_random");
}

int pxl_fill_sphere_points_f64(double* sky, int64_t n, uint64_t seed, uint64_t offset, void* stream) {
    if (n < 0 || (n > 0 && !sky)) return fail(PXL_EINVAL, "fill_sphere: null buffer or negative n");
    if (((uintptr_t)sky & 15) != 0) return fail(PXL_EINVAL, "fill_sphere: 2xN buffer must be 16-byte aligned");
    if (n == 0) return PXL_OK;
    hipLaunchKernelGGL(k_fill_sphere, dim3(stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, (double2*)sky,
                       n, seed, offset);
    return check_launch("k_fill_sphere");
}

}  // extern "C"
