// pxl_maps.h -- whole-map writers: posmap and pixareamap; included by pxl_kernels.hip (one translation unit, -ffp-contract=off).
#pragma once

// ------------------------------------------------------------------------------------------------
// posmap (A15) / pixareamap (N4): write-only maps.  Lane = 2 adjacent RA pixels (16 B stores).
// ------------------------------------------------------------------------------------------------
// Block = (512-column chunk, chunk of rows): RA (rewound) is computed once per lane and reused for every
// row of the chunk; DEC / the row area is one evaluation per row.
#define PXL_POS_ROWS 32
// `fronts`: the row chunks are dealt so that chunk row y works in part y % fronts of each map -- several write fronts per map
// instead of one (a write-only stream takes them faster: see k_pixareamap_chunks).
__global__ __launch_bounds__(256) void k_posmap_car(CarAffine c, int64_t nx, int64_t row0, int64_t nrows,
                                                    double* __restrict__ ra, double* __restrict__ dec, int safe, int fronts) {
    const bool vec = ((nx & 1) == 0) && ((((uintptr_t)ra | (uintptr_t)dec) & 15) == 0);
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;        // 0-based column of the pair
    if (i >= nx) return;
    double a0 = p2s_ra(c, (double)(i + 1));
    double a1 = p2s_ra(c, (double)(i + 2));
    if (safe) { a0 = rewind(a0, PXL_TWOPI_D, 0.0); a1 = rewind(a1, PXL_TWOPI_D, 0.0); }
    const int64_t per = (gridDim.y + fronts - 1) / fronts;
    const int64_t yy = fronts > 1 ? (int64_t)(blockIdx.y % fronts) * per + blockIdx.y / fronts : (int64_t)blockIdx.y;
    const int64_t jr0 = yy * PXL_POS_ROWS;
    if (jr0 >= nrows) return;
    const int64_t jr1 = (jr0 + PXL_POS_ROWS < nrows) ? jr0 + PXL_POS_ROWS : nrows;
    for (int64_t jr = jr0; jr < jr1; ++jr) {
        double d = p2s_dec(c, (double)(row0 + jr + 1));
        if (safe) d = rewind(d, PXL_TWOPI_D, 0.0);
        int64_t o = jr * nx + i;
        if (vec) {
            *reinterpret_cast<double2*>(ra + o) = make_double2(a0, a1);
            *reinterpret_cast<double2*>(dec + o) = make_double2(d, d);
        } else {
            ra[o] = a0; dec[o] = d;
            if (i + 1 < nx) { ra[o + 1] = a1; dec[o + 1] = d; }
        }
    }
}

__global__ __launch_bounds__(256) void k_pixareamap_car(CarAffine c, int64_t nx, int64_t row0, int64_t nrows,
                                                        double* __restrict__ area) {
    const bool vec = ((nx & 1) == 0) && (((uintptr_t)area & 15) == 0);
    const double da = fabs(c.da);
    // one row per blockIdx.y; the row value is computed once per lane and streamed along RA
    const int64_t jr = blockIdx.y;
    const double row = (double)(row0 + jr + 1);
    // enmap_ops.jl:131-134: dec of the two pixel edges, sorted, clamped to the poles
    double e0 = p2s_dec(c, row - 0.5), e1 = p2s_dec(c, row + 0.5);
    double d1 = fmin(e0, e1), d2 = fmax(e0, e1);
    d1 = fmax(-PXL_PI_D / 2, d1); d2 = fmin(PXL_PI_D / 2, d2);
    const double v = (sin(d2) - sin(d1)) * da;
    const int64_t npair = (nx + 1) / 2;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < npair; t += (int64_t)gridDim.x * blockDim.x) {
        int64_t i = t * 2;
        int64_t o = jr * nx + i;
        if (vec) *reinterpret_cast<double2*>(area + o) = make_double2(v, v);
        else { area[o] = v; if (i + 1 < nx) area[o + 1] = v; }
    }
}

// pixareamap, contiguous form: the (nx, nrows) map is one 1-D stream of nx * nrows / 2 pairs; every block writes ONE
// contiguous chunk of PXL_AREA_CHUNK pairs (it spans at most a few rows: the row value is recomputed when the row
// changes).  Same idea as the 2xN evaluators (one contiguous chunk per block); needs nx even and a 16-byte aligned map.
#define PXL_AREA_CHUNK 4096
__device__ inline double pixarea_row(const CarAffine& c, double da, int64_t row0, int64_t jr) {
    const double row = (double)(row0 + jr + 1);
    double e0 = p2s_dec(c, row - 0.5), e1 = p2s_dec(c, row + 0.5);
    double d1 = fmin(e0, e1), d2 = fmax(e0, e1);
    d1 = fmax(-PXL_PI_D / 2, d1); d2 = fmin(PXL_PI_D / 2, d2);
    return (sin(d2) - sin(d1)) * da;
}
__global__ __launch_bounds__(256) void k_pixareamap_chunks(CarAffine c, int64_t nx, int64_t row0, int64_t nrows,
                                                           double* __restrict__ area, int fronts) {
    const double da = fabs(c.da);
    const int64_t npr = nx / 2, total = npr * nrows;
    // FRONTS write fronts: the chunks are dealt so that block b writes in part b % FRONTS of the map (a write-only stream inside
    // one memory class takes 2-8 fronts 11 % faster than one: profiles/r03_fronts_inside_one_class.jsonl)
    const int64_t nchunks = gridDim.x, per = (nchunks + fronts - 1) / fronts;
    const int64_t cidx = fronts > 1 ? (int64_t)(blockIdx.x % fronts) * per + blockIdx.x / fronts : (int64_t)blockIdx.x;
    const int64_t t0 = cidx * PXL_AREA_CHUNK;
    if (t0 >= total) return;
    int64_t jr = t0 / npr;
    int64_t next = (jr + 1) * npr;                           // first pair of the next row
    double v = pixarea_row(c, da, row0, jr);
    double2* out = reinterpret_cast<double2*>(area);
#pragma unroll 4
    for (int it = 0; it < PXL_AREA_CHUNK / 256; ++it) {
        const int64_t t = t0 + it * 256 + threadIdx.x;
        if (t >= total) break;
        while (t >= next) { ++jr; next += npr; v = pixarea_row(c, da, row0, jr); }
        out[t] = make_double2(v, v);
    }
}
