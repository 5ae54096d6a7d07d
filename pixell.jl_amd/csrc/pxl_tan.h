// pxl_tan.h -- Gnomonic (TAN) evaluators; included by pxl_kernels.hip (one translation unit, -ffp-contract=off).
#pragma once

// ------------------------------------------------------------------------------------------------
// Gnomonic (A16), tan_proj.jl:44-75
// ------------------------------------------------------------------------------------------------
struct TanParams { double scale, unit, a0, d0, sd0, cd0, cpx, cpy; };
static TanParams tan_setup(const pxl_car_wcs& w) {
    TanParams t;
    t.scale = 1.0 / w.cdelt[0];
    t.unit = w.unit;
    t.a0 = w.crval[0] * (PXL_PI_D / 180);   // deg2rad.(wcs.crval), tan_proj.jl:47
    t.d0 = w.crval[1] * (PXL_PI_D / 180);
    t.sd0 = sin(t.d0); t.cd0 = cos(t.d0);
    t.cpx = w.crpix[0]; t.cpy = w.crpix[1];
    return t;
}
__device__ inline void tan_sky2pix(const TanParams& t, double a, double d, double* x, double* y) {
    double A = cos(d) * cos(a - t.a0);
    double F = t.scale / t.unit / (t.sd0 * sin(d) + A * t.cd0);
    double LINE = -F * (t.cd0 * sin(d) - A * t.sd0);
    double SAMPLE = -F * cos(d) * sin(a - t.a0);
    *x = t.cpx - SAMPLE;
    *y = t.cpy - LINE;
}
__device__ inline void tan_pix2sky(const TanParams& t, double i, double j, double* a, double* d) {
    double X = (t.cpx - i) * t.unit / t.scale;
    double Y = (t.cpy - j) * t.unit / t.scale;
    double D = atan(sqrt(X * X + Y * Y));
    double B = atan2(-X, Y);
    double sD = sin(D), cD = cos(D), cB = cos(B);
    double XX = t.sd0 * sD * cB + t.cd0 * cD;
    double YY = sD * sin(B);
    *a = t.a0 + atan2(YY, XX);
    *d = asin(t.sd0 * cD - t.cd0 * sD * cB);
}
__global__ __launch_bounds__(256) void k_sky2pix_tan(TanParams t, int64_t n, const double* __restrict__ ra,
                                                     const double* __restrict__ dec, double* __restrict__ x,
                                                     double* __restrict__ y) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride)
        tan_sky2pix(t, ra[k], dec[k], &x[k], &y[k]);
}
__global__ __launch_bounds__(256) void k_pix2sky_tan(TanParams t, int64_t n, const double* __restrict__ ip,
                                                     const double* __restrict__ jp, double* __restrict__ ra,
                                                     double* __restrict__ dec) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride)
        tan_pix2sky(t, ip[k], jp[k], &ra[k], &dec[k]);
}
__global__ __launch_bounds__(256) void k_posmap_tan(TanParams t, int64_t nx, int64_t row0, int64_t nrows,
                                                    double* __restrict__ ra, double* __restrict__ dec) {
    const int64_t total = nx * nrows;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += stride) {
        int64_t jr = k / nx, i = k - jr * nx;
        tan_pix2sky(t, (double)(i + 1), (double)(row0 + jr + 1), &ra[k], &dec[k]);
    }
}
