// native_sharded.cpp -- a host with no torch and no RCCL binding of its own (what a Julia or C program is): it gets its
// communicator from the library (pxl_comm_unique_id / pxl_comm_init_rank), runs pxl_reproject_sharded_step_f64 over
// it and compares the strip with the unsharded reprojection of the same map, bit for bit.  One GPU, so the exchange is a
// loopback (a single-rank communicator; the rank sends rows it owns to itself into its halo slots; the map is built
// so that this delivers exactly what the neighbours would have sent).
//   hipcc --offload-arch=gfx950 -I include tools/native/native_sharded.cpp -L pixell.jl_amd -lpixell_hip -o native_sharded
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "pixell_hip.h"

#define CHECK(x) do { int rc_ = (x); if (rc_ != 0) { char m[512]; pxl_last_error(m, sizeof m); printf("FAIL %s -> %d: %s\n", #x, rc_, m); return 1; } } while (0)
#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAIL %s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    const int64_t nx = 600, ny = 120, nc = 2, nxo = 1200, nyo = 240;
    pxl_car_wcs win = {{-0.6, 0.5}, {300.5, 60.0}, {0.3, 0.0}, 0.017453292519943295};
    pxl_car_wcs wout = {{-0.3, 0.25}, {600.25, 120.3}, {0.3, 0.0}, 0.017453292519943295};
    const int64_t shape_in[3] = {nx, ny, nc}, shape_out[2] = {nxo, nyo};
    const int64_t d0 = 80, dn = 80, own_lo = 41, own_hi = 78;

    unsigned char id[PXL_COMM_ID_BYTES];
    void* comm = nullptr;
    CHECK(pxl_comm_unique_id(id));
    CHECK(pxl_comm_init_rank(id, 0, 1, &comm));
    printf("RCCL instance: %s\n", pxl_comm_backend());

    pxl_reproject_plan *full = nullptr, *strip = nullptr;
    CHECK(pxl_reproject_plan_create(&win, shape_in, 0, ny, &wout, shape_out, 0, nyo, &full));
    int64_t s_lo = 0, s_hi = 0;
    {   // which source rows does the output strip read?
        pxl_reproject_plan* probe = nullptr;
        CHECK(pxl_reproject_plan_create(&win, shape_in, 0, ny, &wout, shape_out, d0, dn, &probe));
        CHECK(pxl_reproject_plan_src_rows(probe, &s_lo, &s_hi));
        pxl_reproject_plan_destroy(probe);
    }
    if (!(s_lo < own_lo && s_hi > own_hi)) { printf("FAIL: the strip needs halo rows on both sides\n"); return 1; }
    CHECK(pxl_reproject_plan_create(&win, shape_in, s_lo, s_hi - s_lo, &wout, shape_out, d0, dn, &strip));

    // the map: pseudo-random, with the halo rows made copies of owned rows (so a self-exchange delivers them)
    std::vector<double> m((size_t)nc * ny * nx);
    unsigned long long z = 88172645463325252ull;
    for (auto& v : m) { z ^= z << 13; z ^= z >> 7; z ^= z << 17; v = (double)(z >> 11) / 9007199254740992.0 - 0.5; }
    const int64_t nbelow = own_lo - s_lo, nabove = s_hi - own_hi;
    for (int64_t c = 0; c < nc; ++c) {
        for (int64_t k = 0; k < nbelow; ++k) memcpy(&m[(c * ny + s_lo + k) * nx], &m[(c * ny + own_lo + 3 + k) * nx], nx * 8);
        for (int64_t k = 0; k < nabove; ++k) memcpy(&m[(c * ny + own_hi + k) * nx], &m[(c * ny + own_lo + 13 + k) * nx], nx * 8);
    }
    double *d_full, *d_out_full, *d_res, *d_out;
    const int64_t rrows = s_hi - s_lo;
    HIPCHECK(hipMalloc(&d_full, m.size() * 8));
    HIPCHECK(hipMalloc(&d_out_full, (size_t)nc * nyo * nxo * 8));
    HIPCHECK(hipMalloc(&d_res, (size_t)nc * rrows * nx * 8));
    HIPCHECK(hipMalloc(&d_out, (size_t)nc * dn * nxo * 8));
    HIPCHECK(hipMemcpy(d_full, m.data(), m.size() * 8, hipMemcpyHostToDevice));
    CHECK(pxl_reproject_execute(full, d_full, d_out_full, nullptr));
    // resident buffer: rows [s_lo, s_hi), halo slots poisoned with NaN
    std::vector<double> res((size_t)nc * rrows * nx);
    for (int64_t c = 0; c < nc; ++c)
        for (int64_t r = 0; r < rrows; ++r) {
            const bool owned = (s_lo + r >= own_lo && s_lo + r < own_hi);
            for (int64_t i = 0; i < nx; ++i) res[(c * rrows + r) * nx + i] = owned ? m[(c * ny + s_lo + r) * nx + i] : NAN;
        }
    HIPCHECK(hipMemcpy(d_res, res.data(), res.size() * 8, hipMemcpyHostToDevice));
    pxl_halo_xfer sends[2] = {{0, 0, own_lo + 3, nbelow}, {0, 0, own_lo + 13, nabove}};
    pxl_halo_xfer recvs[2] = {{0, 0, s_lo, nbelow}, {0, 0, own_hi, nabove}};
    for (int rep = 0; rep < 3; ++rep)
        CHECK(pxl_reproject_sharded_step_f64(strip, d_res, d_out, own_lo, own_hi - own_lo, sends, 2, recvs, 2, comm, nullptr));
    HIPCHECK(hipDeviceSynchronize());
    std::vector<double> a((size_t)nc * dn * nxo), b((size_t)nc * nyo * nxo);
    HIPCHECK(hipMemcpy(a.data(), d_out, a.size() * 8, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(b.data(), d_out_full, b.size() * 8, hipMemcpyDeviceToHost));
    for (int64_t c = 0; c < nc; ++c)
        if (memcmp(&a[c * dn * nxo], &b[(c * nyo + d0) * nxo], (size_t)dn * nxo * 8) != 0) { printf("FAIL: sharded strip differs from the unsharded map (component %lld)\n", (long long)c); return 1; }
    // a communicator the library did not create is refused when the RCCL instance is the library's own
    int rc = pxl_reproject_sharded_step_f64(strip, d_res, d_out, own_lo, own_hi - own_lo, sends, 2, recvs, 2, (void*)0x1234, nullptr);
    if (rc == 0) { printf("FAIL: a foreign communicator was accepted\n"); return 1; }
    if (pxl_comm_destroy((void*)0x1234) == 0) { printf("FAIL: destroying a foreign communicator succeeded\n"); return 1; }
    CHECK(pxl_comm_destroy(comm));
    pxl_reproject_plan_destroy(full);
    pxl_reproject_plan_destroy(strip);
    printf("native_sharded ok: strip of %lld rows bit-identical to the unsharded map over a library-made communicator\n", (long long)dn);
    return 0;
}
