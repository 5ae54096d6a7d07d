/*
 * pixell_hip.h -- C ABI of libpixell_hip.so: MI355X (gfx950) kernels for the Pixell.jl CAR
 * pixel<->sky hot path (pix2sky / sky2pix evaluators, posmap, CAR->CAR bilinear reprojection and
 * scattered bilinear sampling, Float64).
 *
 * This is the drop-in boundary.  The reference (pure Julia) has no plugin ABI of its own; each entry
 * below names the reference method it replaces (file:line under /root/reference/src/).  A Julia host
 * binds these with `ccall((:sym, libpixell_hip), Cint, (...), ...)` in the style the reference already
 * uses for libsharp/wcslib (transforms.jl:185-194, arbitrary_wcs.jl:41-43); see INTEGRATION.md.
 *
 * Conventions
 *   - Plain C types only.  All data pointers are DEVICE pointers (HBM) unless named host_*.
 *   - Every entry returns 0 on success or a negative code (PXL_E*); it never throws, exits or prints.
 *     pxl_last_error() returns the thread-local message of the last failing call.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Entries only enqueue work;
 *     they never synchronise, allocate or free caller-visible memory (plans own their own tables).
 *   - Arrays are Julia column-major: maps are (nx, ny[, nc]) with RA the contiguous axis; coordinate
 *     batches are 2xN, i.e. interleaved (c1, c2) pairs (car_proj.jl:102-107).  Pixel coordinates are
 *     1-based Float64, angles are radians.
 *   - Arithmetic is IEEE double with NO fma contraction, written op-for-op from the reference's source.
 *     What is TESTED: results are bit-identical to the CPU oracle (oracle/pixell_oracle.c); the oracle matches
 *     every literal / data file of the reference's own tests at the reference's own tolerances (isapprox,
 *     100 eps).  The reference itself (Julia) has never been executed next to this library: DESIGN.md 7.
 */
#ifndef PIXELL_HIP_H
#define PIXELL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PXL_VERSION 100   /* 0.1.0 */

/* CarClenshawCurtis{Float64} / CarFejer1{Float64}: projections/car_proj.jl:7-19 (isbits, 56 bytes).
 * Also used for Gnomonic{Float64} (projections/tan_proj.jl:4-9), which has the same fields. */
typedef struct pxl_car_wcs {
    double cdelt[2];
    double crpix[2];
    double crval[2];
    double unit;      /* conversion factor to radians; pi/180 for degree WCS (enmap_geom.jl:18) */
} pxl_car_wcs;

/* error codes */
#define PXL_OK         0
#define PXL_EINVAL   (-22)   /* bad argument (null pointer, negative size, window outside the map) */
#define PXL_ENOMEM   (-12)   /* device allocation failed (plan creation only) */
#define PXL_EHIP     (-5)    /* a HIP runtime call failed; message has hipGetErrorString */
#define PXL_ENODEV   (-19)   /* no gfx950 device / kernel image not loadable */

/* `safe` keyword of the reference's array evaluators */
#define PXL_WRAP_NONE    0   /* safe=false */
#define PXL_WRAP_REWIND  1   /* per-element rewind (what scalar pix2sky does, car_proj.jl:148-150) */
#define PXL_WRAP_UNWIND  2   /* safe=true on 2xN arrays: rewind then unwrap along N (car_proj.jl:110-112) */

/* the reference's three sky2pix roundings (they differ in the last bit) */
#define PXL_FORM_RECIP     0 /* sky2pix!(2xN):      i0 + (a-a0)*(1/d), period abs(2pi/d)      car_proj.jl:165-193 */
#define PXL_FORM_DIV       1 /* sky2pix(ra, dec):   i0 + (a-a0)/d,     period abs(2pi/d)      car_proj.jl:220-234 */
#define PXL_FORM_RECIP_AV  2 /* sky2pix(ras, decs): i0 + (a-a0)*(1/d), period abs(2pi*(1/d))  car_proj.jl:235-252 */

int         pxl_version(void);
/* copies the calling thread's last error message (NUL-terminated, truncated to n) and returns its length */
size_t      pxl_last_error(char* buf, size_t n);
/* number of visible HIP devices, or a negative error */
int         pxl_device_count(void);
/* unwind (safe=true on 2xN batches) takes its scratch (10 bytes per coordinate) from a library-owned stream-ordered
 * pool on the current device and keeps it between calls; this returns the unused part to the driver */
int         pxl_release_scratch(void);

/* ---- pix2sky!(shape, wcs::AbstractCARWCS, pixcoords::2xN, skycoords::2xN; safe)   car_proj.jl:92-115
 *      In-place (pix == sky) is allowed; partially overlapping buffers are refused (PXL_EINVAL) in _UNWIND mode.
 *      wrap_mode: PXL_WRAP_NONE | _REWIND | _UNWIND.  _UNWIND handles at most 2^31-1 points per call.     */
int pxl_pix2sky_car_f64(const pxl_car_wcs* wcs, int64_t n, const double* pix2xN, double* sky2xN,
                        int wrap_mode, void* stream);

/* ---- rewind!(angles; period, ref_angle)                                            enmap_ops.jl:15-19
 *      unwind!(angles; dims, period, ref_angle)                                      enmap_ops.jl:26-32
 *      rewind is elementwise over n doubles.  unwind works along the point axis of an nrow x N column-major
 *      array (nrow = 2: a 2xN coordinate batch with dims=2; nrow = 1: a plain vector), in place.           */
int pxl_rewind_f64(double* a, int64_t n, double period, double ref_angle, void* stream);
int pxl_unwind_f64(double* a, int64_t n, int nrow, double period, double ref_angle, void* stream);

/* ---- pix2sky(shape, wcs, ra_pixel, dec_pixel; safe) broadcast over two N-vectors   car_proj.jl:141-152
 *      safe != 0 -> rewind(ra), rewind(dec).                                                         */
int pxl_pix2sky_car_soa_f64(const pxl_car_wcs* wcs, int64_t n, const double* ipix, const double* jpix,
                            double* ra, double* dec, int safe, void* stream);

/* ---- sky2pix!(shape, wcs, skycoords::2xN, pixcoords::2xN; safe)                    car_proj.jl:165-193
 *      form = PXL_FORM_RECIP reproduces it bit-for-bit; the other forms are exposed for completeness. */
int pxl_sky2pix_car_f64(const pxl_car_wcs* wcs, const int64_t shape[2], int64_t n, const double* sky2xN,
                        double* pix2xN, int safe, int form, void* stream);

/* ---- sky2pix(shape, wcs, ra::AV, dec::AV; safe)   (form = PXL_FORM_RECIP_AV)        car_proj.jl:235-252
 *      sky2pix(shape, wcs, ra::Number, dec::Number) broadcast (form = PXL_FORM_DIV)   car_proj.jl:220-234 */
int pxl_sky2pix_car_soa_f64(const pxl_car_wcs* wcs, const int64_t shape[2], int64_t n, const double* ra,
                            const double* dec, double* ipix, double* jpix, int safe, int form, void* stream);

/* ---- posmap(shape, wcs)                                                             enmap_ops.jl:190-203
 *      Writes rows [row0, row0+nrows) (0-based) of the (nx, ny) RA and DEC maps; ra/dec each hold
 *      nx*nrows doubles.  The reference always uses safe=true (scalar pix2sky -> rewind).             */
int pxl_posmap_car_f64(const pxl_car_wcs* wcs, const int64_t shape[2], int64_t row0, int64_t nrows,
                       double* ra, double* dec, int safe, void* stream);

/* ---- pixareamap!(pixareas)                                                          enmap_ops.jl:124-138
 *      Fills rows [row0, row0+nrows) of an (nx, ny) map with the per-row pixel area (steradians).     */
int pxl_pixareamap_car_f64(const pxl_car_wcs* wcs, const int64_t shape[2], int64_t row0, int64_t nrows,
                           double* area, void* stream);

/* ---- Gnomonic evaluators over N-vectors                                             tan_proj.jl:44-75
 *      (the reference has scalar methods only; posmap(shape, ::Gnomonic) loops them).                 */
int pxl_sky2pix_tan_f64(const pxl_car_wcs* wcs, int64_t n, const double* ra, const double* dec,
                        double* ipix, double* jpix, void* stream);
int pxl_pix2sky_tan_f64(const pxl_car_wcs* wcs, int64_t n, const double* ipix, const double* jpix,
                        double* ra, double* dec, void* stream);
int pxl_posmap_tan_f64(const pxl_car_wcs* wcs, const int64_t shape[2], int64_t row0, int64_t nrows,
                       double* ra, double* dec, void* stream);

/* ---- Bilinear reprojection CAR -> CAR.  NOT in the reference: the composite
 *      posmap(out) [enmap_ops.jl:190-203, safe=false] o sky2pix(in) [car_proj.jl:220-234, safe=true]
 *      o 2x2 gather + lerp, defined by oracle/pixell_oracle.c.
 *
 *      A plan holds the separable coordinate tables (nx_out + ny_out entries) in device memory.
 *      src holds rows [src_row0, src_row0+src_nrows) of every component plane of the (nx, ny, nc)
 *      source map: (nx, src_nrows, nc) column-major; dst receives rows [dst_row0, dst_row0+dst_nrows)
 *      of the (nx_out, ny_out, nc) output: (nx_out, dst_nrows, nc).  Full maps: row0 = 0, nrows = ny.
 *      Windows are how a declination strip (+ halo rows) of a sharded map is described without
 *      changing a single bit of the coordinate arithmetic.                                          */
typedef struct pxl_reproject_plan pxl_reproject_plan;

int pxl_reproject_plan_create(const pxl_car_wcs* wcs_in, const int64_t shape_in[3],
                              int64_t src_row0, int64_t src_nrows,
                              const pxl_car_wcs* wcs_out, const int64_t shape_out[2],
                              int64_t dst_row0, int64_t dst_nrows,
                              pxl_reproject_plan** plan);
/* enqueue table build + reprojection of all nc components on `stream` */
int pxl_reproject_execute(pxl_reproject_plan* plan, const double* src, double* dst, void* stream);
/* as above but only output rows [r0, r0+nr) RELATIVE to the plan's dst window (for interior/boundary
 * splitting while a halo is in flight); the tables must have been built by a previous execute/build ON THE
 * SAME STREAM (or one the caller has ordered before this one).  A plan may be shared by host threads only
 * for concurrent execute_rows calls; create/build/destroy are the owner's.  src and dst must not overlap.
 * Tuning knobs read once at plan creation: PXL_REPROJECT_{RH,PAIRS,NS,PF,VARIANT,FLAGS}.   */
int pxl_reproject_build_tables(pxl_reproject_plan* plan, void* stream);
int pxl_reproject_execute_rows(pxl_reproject_plan* plan, const double* src, double* dst,
                               int64_t r0, int64_t nr, void* stream);
/* Float32 maps (Enmap{Float32}: the storage type most released maps use).  Same plan, same Float64 coordinate
 * tables and weights; taps are widened to Float64, the result is rounded once to Float32 -- what Julia does when
 * a Float64 expression is assigned into a Float32 array.                                                        */
int pxl_reproject_execute_f32(pxl_reproject_plan* plan, const float* src, float* dst, void* stream);
int pxl_reproject_execute_rows_f32(pxl_reproject_plan* plan, const float* src, float* dst,
                                   int64_t r0, int64_t nr, void* stream);
/* source rows [lo, hi) (0-based, absolute) that the plan's dst window reads (host computation,
 * same arithmetic as the device tables) -- what a shard must hold, i.e. strip + halo.               */
int pxl_reproject_plan_src_rows(const pxl_reproject_plan* plan, int64_t* lo, int64_t* hi);
/* output rows (relative to the dst window, [lo, hi)) whose stencil lies entirely inside source rows
 * [have_lo, have_hi) -- the interior that can start before a halo arrives.                           */
int pxl_reproject_plan_rows_covered(const pxl_reproject_plan* plan, int64_t have_lo, int64_t have_hi,
                                    int64_t* lo, int64_t* hi);
/* tuning knob for benchmarking / cross-checks: 0 = auto (LDS-DMA kernel when the source allows 16-B loads),
 * 1 = direct-gather kernel, 2 = register-staged kernel */
int pxl_reproject_plan_set_variant(pxl_reproject_plan* plan, int variant);
int pxl_reproject_plan_destroy(pxl_reproject_plan* plan);

/* ---- One step of the dec-strip sharded operator on this rank: exchange halo rows with the
 *      neighbouring ranks over RCCL send/recv, reproject the output rows that need only owned rows while the
 *      halo travels, then the rest.  The plan describes this rank's windows: src is its resident buffer
 *      ([nc,] src_nrows, nx) holding the rows it owns -- [own_row0, own_row0 + own_nrows), absolute -- with room
 *      for the halo rows around them; sends/recvs list absolute source rows (one RCCL message per component plane,
 *      straight from/into src: nothing is staged).  rccl_comm is the caller's ncclComm_t (one rank per GPU) or one made
 *      by pxl_comm_init_rank; the RCCL functions are looked up in the RCCL instance already loaded in the process
 *      (PXL_RCCL_LIB names it explicitly), never in a second instance loaded behind the caller's back, so this
 *      library has no link-time RCCL dependency.  The exchange runs on a stream owned by the plan,
 *      ordered after everything queued on `stream` before the call; the function does not synchronise.
 *      With no transfers (one rank) it is build_tables + execute.  No reference counterpart.                    */
typedef struct pxl_halo_xfer {
    int32_t peer;        /* rank in rccl_comm */
    int32_t reserved;
    int64_t row0;        /* first source row of the transfer, 0-based, absolute */
    int64_t nrows;
} pxl_halo_xfer;
int pxl_reproject_sharded_step_f64(pxl_reproject_plan* plan, double* src, double* dst,
                                   int64_t own_row0, int64_t own_nrows,
                                   const pxl_halo_xfer* sends, int nsends, const pxl_halo_xfer* recvs, int nrecvs,
                                   void* rccl_comm, void* stream);
int pxl_reproject_sharded_step_f32(pxl_reproject_plan* plan, float* src, float* dst,
                                   int64_t own_row0, int64_t own_nrows,
                                   const pxl_halo_xfer* sends, int nsends, const pxl_halo_xfer* recvs, int nrecvs,
                                   void* rccl_comm, void* stream);

/* ---- RCCL communicator for hosts that have no RCCL binding of their own (a Julia or C host; torch users pass
 *      ProcessGroupNCCL's communicator instead).  Thin wrappers over ncclGetUniqueId / ncclCommInitRank /
 *      ncclCommDestroy.  One rank calls pxl_comm_unique_id and distributes the PXL_COMM_ID_BYTES bytes to the others by
 *      any means (file, MPI, socket); every rank then calls pxl_comm_init_rank (collective) with its HIP device current.
 *      RCCL is looked up in the instance already loaded in the process, else PXL_RCCL_LIB, else the library loads
 *      librccl.so itself -- and then pxl_reproject_sharded_step_* only accepts communicators created here (a
 *      communicator means nothing to another RCCL instance).  pxl_comm_backend() says which instance is in use.     */
#define PXL_COMM_ID_BYTES 128
int pxl_comm_unique_id(void* id128);
int pxl_comm_init_rank(const void* id128, int rank, int nranks, void** comm);
int pxl_comm_destroy(void* comm);
const char* pxl_comm_backend(void);

/* one-shot convenience (creates a plan, executes, synchronises `stream`, destroys) */
int pxl_reproject_car_bilinear_f64(const pxl_car_wcs* wcs_in, const int64_t shape_in[3], const double* src,
                                   const pxl_car_wcs* wcs_out, const int64_t shape_out[2], double* dst,
                                   void* stream);

/* ---- Generic (non-separable) bilinear reprojection between CAR and Gnomonic maps: per output pixel
 *      pix2sky(out) -> sky2pix(in) -> 2x2 gather, with the evaluators of car_proj.jl / tan_proj.jl.
 *      The coordinate map is interpolated per 128 x 32 output tile (degree 6 x 5 from 42 exact evaluations) and the
 *      interpolant checked against twelve more exact evaluations to 1e-10 pixel; tiles that fail the check (and every tile with
 *      PXL_GENERIC_EXACT=1) evaluate per pixel (~10 FP64 libm calls each).  Full maps only.
 *      proj codes: PXL_PROJ_CAR, PXL_PROJ_TAN.                                                                   */
#define PXL_PROJ_CAR 0
#define PXL_PROJ_TAN 1
int pxl_reproject_generic_bilinear_f64(const pxl_car_wcs* wcs_in, int proj_in, const int64_t shape_in[3],
                                       const double* src, const pxl_car_wcs* wcs_out, int proj_out,
                                       const int64_t shape_out[2], double* dst, void* stream);

/* The same operator with its coordinate lattice kept: what pxl_reproject_plan is to the separable CAR -> CAR path.  Creating
 * the plan evaluates every tile's lattice and check points once (k_generic_lattice) and records which tiles must be evaluated
 * per pixel; executing it is the pixel kernel alone (plus the per-pixel launch only when such tiles exist), on any source map
 * of the plan's input geometry with any number of components.  A mosaic of patches, or many maps onto one patch, pays the
 * transcendental part once per geometry pair.  Same results as the one-shot entry, bit for bit (same kernels, same lattice).
 * create synchronises `stream` (it reads the count of per-pixel tiles back); execute is asynchronous on its stream.
 * 2.8 MB of device memory per 4096 x 4096 output patch (676 B per tile), owned by the plan.                           */
typedef struct pxl_generic_plan pxl_generic_plan;
int pxl_generic_plan_create(const pxl_car_wcs* wcs_in, int proj_in, const int64_t shape_in[2],
                            const pxl_car_wcs* wcs_out, int proj_out, const int64_t shape_out[2],
                            void* stream, pxl_generic_plan** plan);
int pxl_generic_plan_execute(const pxl_generic_plan* plan, int64_t ncomp, const double* src, double* dst, void* stream);
int pxl_generic_plan_tiles(const pxl_generic_plan* plan, int64_t* exact_tiles, int64_t* total_tiles);
int pxl_generic_plan_destroy(pxl_generic_plan* plan);

/* diagnostics: of the 128 x 32 output tiles of the last pxl_reproject_generic_bilinear_f64 call on the current device,
 * how many evaluated the coordinates per pixel (the interpolant failed its 1e-10-pixel check there: the rewind jump of
 * a periodic source, the Gnomonic horizon, very coarse pixels).  Synchronises `stream`.                          */
int pxl_reproject_generic_last_tiles(int64_t* exact_tiles, int64_t* total_tiles, void* stream);

/* ---- Scattered bilinear sample: (x, y) = sky2pix!(shape_in, wcs_in, sky2xN; safe=true)
 *      [car_proj.jl:165-193] then the same 2x2 gather + lerp.  out is (n, nc) column-major.          */
int pxl_sample_car_bilinear_f64(const pxl_car_wcs* wcs_in, const int64_t shape_in[3], const double* src,
                                int64_t src_row0, int64_t src_nrows,
                                int64_t n, const double* sky2xN, double* out, void* stream);

int pxl_sample_car_bilinear_f32(const pxl_car_wcs* wcs_in, const int64_t shape_in[3], const float* src,
                                int64_t src_row0, int64_t src_nrows,
                                int64_t n, const double* sky2xN, float* out, void* stream);

/* ---- The same sample from a ROW-PAIR copy of the map (caller-owned, 64-byte aligned, pxl_sample_pairs_elems() map
 *      elements = 8/3 of the resident window plus one row): an entry holds (v[p-1][i], v[p][i]), and the entries of a
 *      row are stored in 64-byte groups that overlap by one entry (4 Float64 / 8 Float32 entries per group, columns
 *      past nx wrapping round), so a point's whole 2x2 neighbourhood is two adjacent entries of ONE 64-byte sector:
 *      1.0 instead of 2.25 random sectors per point.  The element count covers either element type (a Float32 copy
 *      of a large map uses 6/7 of it).  Build once per map (one streaming pass), sample any number of batches; results are
 *      bit-identical to pxl_sample_car_bilinear_*.  No reference counterpart (the reference has no sampler,
 *      SURVEY 8(a) R1).                                                                                          */
int64_t pxl_sample_pairs_elems(const int64_t shape_in[3], int64_t src_nrows);      /* -1 on invalid arguments */
int pxl_sample_build_pairs_f64(const int64_t shape_in[3], const double* src, int64_t src_nrows, double* pairs,
                               void* stream);
int pxl_sample_build_pairs_f32(const int64_t shape_in[3], const float* src, int64_t src_nrows, float* pairs,
                               void* stream);
int pxl_sample_car_bilinear_pairs_f64(const pxl_car_wcs* wcs_in, const int64_t shape_in[3], const double* pairs,
                                      int64_t src_row0, int64_t src_nrows,
                                      int64_t n, const double* sky2xN, double* out, void* stream);
int pxl_sample_car_bilinear_pairs_f32(const pxl_car_wcs* wcs_in, const int64_t shape_in[3], const float* pairs,
                                      int64_t src_row0, int64_t src_nrows,
                                      int64_t n, const double* sky2xN, float* out, void* stream);

/* ---- FITS image staging (the on-disk format either side of the path: read_map / write_map, enmap.jl:198-237).
 *      raw_be: device copy of the HDU's big-endian data block, n elements of BITPIX -64 (or -32 for decode);
 *      decode writes native Float64 (in place allowed for -64), encode writes big-endian Float64.          */
int pxl_fits_decode_f64(const void* raw_be, double* dst, int64_t n, int bitpix, void* stream);
int pxl_fits_encode_f64(const double* src, void* raw_be, int64_t n, void* stream);
/* BITPIX -32 <-> native Float32 (the 4-byte swap is its own inverse; in place allowed) */
int pxl_fits_swap_f32(const void* src, void* dst, int64_t n, void* stream);

/* ---- Placement probe.  The memory of a hipMalloc'ed allocation on the MI355X falls into three classes (thirds of the 288 GiB:
 *      DESIGN.md 4.7, profiles/r03_xcd_classes.txt): a kernel with several far-apart WRITE fronts -- the reprojection keeps one
 *      per XCD -- stores at 5.8-6.0 TB/s when all of them lie in one class and at 6.8-7.1 TB/s when they are split over two.
 *      This entry times the pattern that defines the classes: eight store fronts (one per XCD), four in window a and four in
 *      window b, each writing window_bytes / 4 bytes of ZEROS (both windows are overwritten).  `us` receives the median of
 *      `reps` launches in microseconds; with 1 GiB windows about 380 us = same class, 305 us = different classes.  It
 *      synchronises `stream`.  A host can map an allocation with it and put a destination across a class boundary
 *      (pixell.jl_amd/placement.py does; pxl_mem_pair_alloc below is the same rule natively).                            */
int pxl_mem_probe_pair(void* a, void* b, size_t window_bytes, int reps, float* us, void* stream);

/* A (source, destination) pair of maps placed by that rule: ONE hipMalloc of src + dst + headroom bytes (capped at the free
 * memory less 6 GiB), its classes mapped with the probe (one 1 GiB window every 2 GiB), the destination centred on the
 * boundary between two classes that has the most room on both sides, the source in a stretch of a class the destination does
 * not touch -- looked for in separate allocations (up to 96 GiB of candidates, freed again) when the allocation holds none.
 * Without a boundary inside the allocation the layout is the plain one (source first, destination at the top).  The source is
 * zero-filled; both pointers are 2 MiB aligned.  Topology discovery only: nothing about the caller's kernel is timed.  The
 * head-room stays allocated until pxl_mem_pair_free (hipMalloc cannot return part of an allocation): 144 GiB is what it takes
 * for all three classes to show up on a fresh device.  Uses the current device; synchronises `stream`.                    */
typedef struct pxl_mem_pair {
    void* src;                 /* src_bytes, zero-filled */
    void* dst;                 /* dst_bytes */
    void* arena;               /* the allocation dst (and normally src) lives in */
    void* src_alloc;           /* the separate allocation holding src, or NULL */
    uint64_t arena_bytes;
    uint64_t src_offset;       /* of src within arena (0 when src_alloc is set) */
    uint64_t dst_offset;
    int32_t classes;           /* labels given to the allocation's windows: the part has 3 classes; a window that straddles a
                                  boundary can get a label of its own */
    int32_t dst_two_classes;   /* 1: dst straddles a class boundary */
    int32_t src_own_class;     /* 1: src lies in a class dst does not touch */
    int32_t probes;            /* probe launches spent (0.3 ms each) */
    int32_t separate_tried;    /* separate source allocations tried */
    int32_t reserved_;
} pxl_mem_pair;
int pxl_mem_pair_alloc(uint64_t src_bytes, uint64_t dst_bytes, uint64_t headroom_bytes, pxl_mem_pair* out, void* stream);
int pxl_mem_pair_free(pxl_mem_pair* pair);

/* The library's DEFAULT allocation policy for a map-sized output (round 4; the reference's seam: `similar` keeps the array type,
 * src/enmap.jl:60-62, so a device host allocates its outputs through its own allocator -- this one).  ONE buffer of `bytes`
 * bytes that lies in TWO memory classes, with NO head-room kept: the buffer is allocated on its own (hipMalloc), its 1 GiB
 * windows are labelled with the probe above, and while it lies inside one class it is held as ballast and the next allocation
 * is tried (consecutive allocations walk through the device's memory; a class run is 4-32 GiB long).  A candidate with at least
 * 30 % of its windows in a second class (20 % for buffers of 16 GiB and more) is taken at once; otherwise the best one seen within `budget_bytes` of ballast (0 = 96
 * GiB; never more than the free memory less 8 GiB) and 24 tries.  All ballast is freed before the call returns.  Buffers
 * below 3 GiB are plain allocations.  The same fixed rule as pixell.jl_amd/placement.py::empty_map (what pj.reproject allocates
 * its output with); topology discovery only, nothing about the caller's kernel is timed.  Contents unspecified (probed windows
 * hold zeros).  Free with pxl_mem_free.  Uses the current device; synchronises `stream`.                                       */
typedef struct pxl_mem_placed_info {
    int32_t tries;             /* allocations made */
    int32_t probes;            /* probe launches spent */
    int32_t two_classes;       /* 1: at least 20 % of the buffer's windows lie in a second class */
    int32_t minor_share_pct;   /* share of the buffer's windows outside its majority class, in percent */
    uint64_t ballast_bytes;    /* largest amount of rejected candidates held at once (all freed on return) */
} pxl_mem_placed_info;
int pxl_mem_alloc_placed(uint64_t bytes, uint64_t budget_bytes, void** out, pxl_mem_placed_info* info, void* stream);
int pxl_mem_free(void* ptr);

/* ---- synthetic inputs (benchmark plumbing, deterministic counter-based generator):
 *      fill n doubles with N(0,1) (kind 0) or U[0,1) (kind 1) from splitmix64(seed, index+offset);
 *      uniform-on-sphere points (ra = 2pi*u1 - pi, dec = asin(2*u2 - 1)) as a 2xN batch.             */
int pxl_fill_random_f64(double* dst, int64_t n, uint64_t seed, uint64_t offset, int kind, void* stream);
int pxl_fill_sphere_points_f64(double* sky2xN, int64_t n, uint64_t seed, uint64_t offset, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PIXELL_HIP_H */
