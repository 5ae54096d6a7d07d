# PixellHIP.jl -- the reference-side binding: new methods of Pixell's own generic functions for maps whose
# storage lives in MI355X HBM, each a thin `ccall` into libpixell_hip.so (include/pixell_hip.h).
#
# NOT EXERCISED IN THIS REPOSITORY'S CI: Julia is not installed in the build image (see DESIGN.md 1); the
# executable twin of this file is the Python host package pixell.jl_amd/.  It follows the reference's own FFI
# style (`ccall((:sym, lib), Cint, (...), ...)` + `GC.@preserve`, src/transforms.jl:185-194) but checks the
# returned code.  No CUDA.jl / AMDGPU.jl: device memory is a minimal array type over hipMalloc.
module PixellHIP

using Pixell
import Pixell: Enmap, AbstractCARWCS, CarClenshawCurtis, Gnomonic, getwcs, pix2sky, pix2sky!, sky2pix, sky2pix!, posmap,
               pixareamap!, rewind, rewind!, unwind, unwind!, read_map, write_map
using WCS: WCSTransform
import WCS
using FITSIO: FITS, read_header
using Printf: @sprintf

const libpixell_hip = get(ENV, "PIXELL_HIP_LIB", "libpixell_hip.so")
const libhip = "libamdhip64.so"

# ---- struct pxl_car_wcs == CarClenshawCurtis{Float64} field for field (car_proj.jl:7-12): pass by Ref
struct CarWCS
    cdelt::NTuple{2,Cdouble}
    crpix::NTuple{2,Cdouble}
    crval::NTuple{2,Cdouble}
    unit::Cdouble
end
CarWCS(w::AbstractCARWCS) = CarWCS(Float64.(w.cdelt), Float64.(w.crpix), Float64.(w.crval), Float64(w.unit))
# Gnomonic{T} has the same four fields (tan_proj.jl:4-9); the C side takes it through the same struct
CarWCS(w::Gnomonic) = CarWCS(Float64.(w.cdelt), Float64.(w.crpix), Float64.(w.crval), Float64(w.unit))
const AnyFastWCS = Union{AbstractCARWCS,Gnomonic}
projcode(::AbstractCARWCS) = Cint(0)                           # PXL_PROJ_CAR
projcode(::Gnomonic) = Cint(1)                                 # PXL_PROJ_TAN

function check(rc::Cint)
    rc == 0 && return nothing
    buf = Vector{UInt8}(undef, 512)
    ccall((:pxl_last_error, libpixell_hip), Csize_t, (Ptr{UInt8}, Csize_t), buf, 512)
    error("libpixell_hip ($rc): " * unsafe_string(pointer(buf)))     # the reference raises exceptions too
end

# ---- the library's allocation policy for device arrays (include/pixell_hip.h: pxl_mem_alloc_placed)
struct MemPlacedInfo                    # == struct pxl_mem_placed_info
    tries::Int32
    probes::Int32
    two_classes::Int32
    minor_share_pct::Int32
    ballast_bytes::UInt64
end
const ALLOC_POLICY = Ref{Symbol}(:class_aware)          # or :plain

# ---- minimal device array (column-major, like Array): the `AA` slot of Enmap{T,N,AA,W} (enmap.jl:10)
mutable struct HIPArray{T,N} <: AbstractArray{T,N}
    ptr::Ptr{T}
    dims::NTuple{N,Int}
    parent::Any
    # Every fresh device array -- `similar` (enmap.jl:60-62 keeps the array type), the outputs of reproject / pix2sky / posmap --
    # comes from the library's allocation policy (pxl_mem_alloc_placed): below 3 GiB a plain hipMalloc; a map-sized buffer is
    # looked for across a boundary between two of the HBM's memory classes, where the reprojection's write fronts store 15 %
    # faster, and nothing but the buffer stays allocated afterwards.  ALLOC_POLICY[] = :plain turns that off.
    function HIPArray{T,N}(::UndefInitializer, dims::NTuple{N,Int}) where {T,N}
        p = Ref{Ptr{Cvoid}}(C_NULL)
        nbytes = max(prod(dims) * sizeof(T), 1)
        if ALLOC_POLICY[] === :plain
            rc = ccall((:hipMalloc, libhip), Cint, (Ptr{Ptr{Cvoid}}, Csize_t), p, nbytes)
            rc == 0 || error("hipMalloc failed ($rc)")
        else
            info = Ref(MemPlacedInfo(0, 0, 0, 0, 0))
            check(ccall((:pxl_mem_alloc_placed, libpixell_hip), Cint, (UInt64, UInt64, Ptr{Ptr{Cvoid}}, Ref{MemPlacedInfo}, Ptr{Cvoid}),
                        nbytes, 0, p, info, C_NULL))
        end
        a = new{T,N}(Ptr{T}(p[]), dims, nothing)
        finalizer(x -> ccall((:pxl_mem_free, libpixell_hip), Cint, (Ptr{Cvoid},), x.ptr), a)      # hipFree either way
    end
    # a view into `parent`'s memory (kept alive by the reference; no finalizer of its own): place_pair below
    HIPArray{T,N}(ptr::Ptr{T}, dims::NTuple{N,Int}, parent) where {T,N} = new{T,N}(ptr, dims, parent)
end
HIPArray{T}(u::UndefInitializer, dims::Int...) where {T} = HIPArray{T,length(dims)}(u, dims)
Base.size(a::HIPArray) = a.dims
Base.similar(a::HIPArray{T}, ::Type{S}, dims::Dims) where {T,S} = HIPArray{S,length(dims)}(undef, dims)
Base.unsafe_convert(::Type{Ptr{T}}, a::HIPArray{T}) where {T} = a.ptr
Base.getindex(::HIPArray, i...) = error("scalar indexing of device memory: copy to the host with Array(a)")
function HIPArray(h::Array{T,N}) where {T,N}
    d = HIPArray{T,N}(undef, size(h))
    GC.@preserve h ccall((:hipMemcpy, libhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Cint), d.ptr, pointer(h), sizeof(h), 1)
    d
end
function Base.Array(d::HIPArray{T,N}) where {T,N}
    h = Array{T,N}(undef, d.dims)
    GC.@preserve h ccall((:hipMemcpy, libhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Cint), pointer(h), d.ptr, sizeof(h), 2)
    h
end

function Base.copy(d::HIPArray{T,N}) where {T,N}            # device-to-device (hipMemcpyDeviceToDevice = 3)
    c = HIPArray{T,N}(undef, d.dims)
    GC.@preserve d c ccall((:hipMemcpy, libhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Cint), c.ptr, d.ptr, prod(d.dims) * sizeof(T), 3)
    c
end

const DevCoords = HIPArray{Float64,2}                      # 2xN, interleaved pairs (car_proj.jl:102-107)
const DevVector = HIPArray{Float64,1}
const NULLSTREAM = C_NULL

# ---- pix2sky! / sky2pix! for 2xN device batches: same signatures and `safe` keyword as car_proj.jl:92,165
function Pixell.pix2sky!(shape, wcs::AbstractCARWCS, pix::DevCoords, sky::DevCoords; safe=true)
    @assert size(pix) == size(sky) && size(pix, 1) == 2
    GC.@preserve pix sky check(ccall((:pxl_pix2sky_car_f64, libpixell_hip), Cint,
        (Ref{CarWCS}, Int64, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Ptr{Cvoid}),
        CarWCS(wcs), size(pix, 2), pix.ptr, sky.ptr, safe ? 2 : 0, NULLSTREAM))   # 2 = PXL_WRAP_UNWIND
    return sky
end
Pixell.pix2sky(shape, wcs::AbstractCARWCS, pix::DevCoords; safe=true) =
    pix2sky!(shape, wcs, pix, similar(pix); safe=safe)

function Pixell.sky2pix!(shape, wcs::AbstractCARWCS, sky::DevCoords, pix::DevCoords; safe=true)
    @assert size(pix) == size(sky) && size(sky, 1) == 2
    shp = Int64[shape[1], shape[2]]
    GC.@preserve pix sky shp check(ccall((:pxl_sky2pix_car_f64, libpixell_hip), Cint,
        (Ref{CarWCS}, Ptr{Int64}, Int64, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint, Ptr{Cvoid}),
        CarWCS(wcs), shp, size(sky, 2), sky.ptr, pix.ptr, safe, 0, NULLSTREAM))    # 0 = PXL_FORM_RECIP
    return pix
end
Pixell.sky2pix(shape, wcs::AbstractCARWCS, sky::DevCoords; safe=true) =
    sky2pix!(shape, wcs, sky, similar(sky); safe=safe)

# ---- pix2sky(shape, wcs, ra_pixel, dec_pixel; safe) on two device N-vectors (car_proj.jl:141-152: the broadcast form;
#      safe -> rewind, as the reference's FIXME leaves it).  Without this method a HIPArray falls into the reference's
#      broadcast and stops at the scalar-indexing error.
function Pixell.pix2sky(shape, wcs::AbstractCARWCS, ra_pixel::DevVector, dec_pixel::DevVector; safe=true)
    @assert length(ra_pixel) == length(dec_pixel)
    ra, dec = similar(ra_pixel), similar(dec_pixel)
    GC.@preserve ra_pixel dec_pixel ra dec check(ccall((:pxl_pix2sky_car_soa_f64, libpixell_hip), Cint,
        (Ref{CarWCS}, Int64, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Ptr{Cvoid}),
        CarWCS(wcs), length(ra_pixel), ra_pixel.ptr, dec_pixel.ptr, ra.ptr, dec.ptr, safe, NULLSTREAM))
    return ra, dec
end

# ---- sky2pix(shape, wcs, ra::AV, dec::AV; safe) on two device N-vectors (car_proj.jl:235-252: reciprocal form with the
#      period abs(2pi * (1/d)), PXL_FORM_RECIP_AV = 2)
function Pixell.sky2pix(shape, wcs::AbstractCARWCS, ra::DevVector, dec::DevVector; safe=true)
    @assert length(ra) == length(dec)
    pix_ra, pix_dec = similar(ra), similar(dec)
    shp = Int64[shape[1], shape[2]]
    GC.@preserve ra dec pix_ra pix_dec shp check(ccall((:pxl_sky2pix_car_soa_f64, libpixell_hip), Cint,
        (Ref{CarWCS}, Ptr{Int64}, Int64, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint, Ptr{Cvoid}),
        CarWCS(wcs), shp, length(ra), ra.ptr, dec.ptr, pix_ra.ptr, pix_dec.ptr, safe, 2, NULLSTREAM))
    return pix_ra, pix_dec
end

# ---- Gnomonic evaluators over device N-vectors (tan_proj.jl:44-75; the reference has the scalar methods only and
#      ignores `safe`, so does the library)
function Pixell.sky2pix(shape, wcs::Gnomonic, ra::DevVector, dec::DevVector; safe=false)
    @assert length(ra) == length(dec)
    x, y = similar(ra), similar(dec)
    GC.@preserve ra dec x y check(ccall((:pxl_sky2pix_tan_f64, libpixell_hip), Cint,
        (Ref{CarWCS}, Int64, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cvoid}),
        CarWCS(wcs), length(ra), ra.ptr, dec.ptr, x.ptr, y.ptr, NULLSTREAM))
    return x, y
end
function Pixell.pix2sky(shape, wcs::Gnomonic, ra_pixel::DevVector, dec_pixel::DevVector; safe=false)
    @assert length(ra_pixel) == length(dec_pixel)
    ra, dec = similar(ra_pixel), similar(dec_pixel)
    GC.@preserve ra_pixel dec_pixel ra dec check(ccall((:pxl_pix2sky_tan_f64, libpixell_hip), Cint,
        (Ref{CarWCS}, Int64, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cvoid}),
        CarWCS(wcs), length(ra_pixel), ra_pixel.ptr, dec_pixel.ptr, ra.ptr, dec.ptr, NULLSTREAM))
    return ra, dec
end
# posmap(shape, wcs::Gnomonic) (enmap_ops.jl:190-203 over tan_proj.jl:59-75) into device maps
function posmap_device(shape::Tuple{Int,Int}, wcs::Gnomonic)
    ra, dec = HIPArray{Float64}(undef, shape...), HIPArray{Float64}(undef, shape...)
    shp = Int64[shape...]
    GC.@preserve ra dec shp check(ccall((:pxl_posmap_tan_f64, libpixell_hip), Cint,
        (Ref{CarWCS}, Ptr{Int64}, Int64, Int64, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cvoid}),
        CarWCS(wcs), shp, 0, shape[2], ra.ptr, dec.ptr, NULLSTREAM))
    return Enmap(ra, wcs), Enmap(dec, wcs)
end

# ---- posmap on the device (enmap_ops.jl:190-203): returns two device Enmaps
function posmap_device(shape::Tuple{Int,Int}, wcs::AbstractCARWCS)
    ra, dec = HIPArray{Float64}(undef, shape...), HIPArray{Float64}(undef, shape...)
    shp = Int64[shape...]
    GC.@preserve ra dec shp check(ccall((:pxl_posmap_car_f64, libpixell_hip), Cint,
        (Ref{CarWCS}, Ptr{Int64}, Int64, Int64, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Ptr{Cvoid}),
        CarWCS(wcs), shp, 0, shape[2], ra.ptr, dec.ptr, 1, NULLSTREAM))
    return Enmap(ra, wcs), Enmap(dec, wcs)
end

# ---- bilinear reprojection of a device Enmap onto (shape_out, wcs_out): the composite of
#      posmap(out) o sky2pix(in) o 2x2 gather (not in the reference; SURVEY 8(a) R1)
function reproject(m::Enmap{Float64,N,<:HIPArray,<:AbstractCARWCS}, shape_out::Tuple{Int,Int},
                   wcs_out::AbstractCARWCS) where {N}
    nc = N == 3 ? size(m, 3) : 1
    out = HIPArray{Float64}(undef, shape_out..., (N == 3 ? (nc,) : ())...)
    shp_in, shp_out = Int64[size(m, 1), size(m, 2), nc], Int64[shape_out...]
    src = parent(m)
    GC.@preserve src out shp_in shp_out check(ccall((:pxl_reproject_car_bilinear_f64, libpixell_hip), Cint,
        (Ref{CarWCS}, Ptr{Int64}, Ptr{Cdouble}, Ref{CarWCS}, Ptr{Int64}, Ptr{Cdouble}, Ptr{Cvoid}),
        CarWCS(getwcs(m)), shp_in, src.ptr, CarWCS(wcs_out), shp_out, out.ptr, NULLSTREAM))
    return Enmap(out, wcs_out)
end

# ---- CAR <-> Gnomonic (and TAN -> TAN): the non-separable reprojection, pix2sky(out) -> sky2pix(in) -> 2x2 gather with
#      the evaluators of car_proj.jl / tan_proj.jl.  Picked by dispatch when either side is a Gnomonic WCS.
function reproject(m::Enmap{Float64,N,<:HIPArray,<:AnyFastWCS}, shape_out::Tuple{Int,Int}, wcs_out::Gnomonic) where {N}
    return reproject_generic(m, shape_out, wcs_out)
end
function reproject(m::Enmap{Float64,N,<:HIPArray,<:Gnomonic}, shape_out::Tuple{Int,Int}, wcs_out::AbstractCARWCS) where {N}
    return reproject_generic(m, shape_out, wcs_out)
end
function reproject_generic(m::Enmap{Float64,N,<:HIPArray}, shape_out::Tuple{Int,Int}, wcs_out::AnyFastWCS) where {N}
    nc = N == 3 ? size(m, 3) : 1
    out = HIPArray{Float64}(undef, shape_out..., (N == 3 ? (nc,) : ())...)
    shp_in, shp_out = Int64[size(m, 1), size(m, 2), nc], Int64[shape_out...]
    src = parent(m)
    win = getwcs(m)
    GC.@preserve src out shp_in shp_out check(ccall((:pxl_reproject_generic_bilinear_f64, libpixell_hip), Cint,
        (Ref{CarWCS}, Cint, Ptr{Int64}, Ptr{Cdouble}, Ref{CarWCS}, Cint, Ptr{Int64}, Ptr{Cdouble}, Ptr{Cvoid}),
        CarWCS(win), projcode(win), shp_in, src.ptr, CarWCS(wcs_out), projcode(wcs_out), shp_out, out.ptr, NULLSTREAM))
    return Enmap(out, wcs_out)
end

# ---- the same with the coordinate lattice kept (pxl_generic_plan_*): make once per pair of geometries, execute per map
mutable struct GenericReprojectPlan
    handle::Ptr{Cvoid}
    shape_in::NTuple{2,Int}
    shape_out::NTuple{2,Int}
    wcs_out::AnyFastWCS
    function GenericReprojectPlan(shape_in, wcs_in::AnyFastWCS, shape_out, wcs_out::AnyFastWCS)
        shp_in, shp_out = Int64[shape_in[1], shape_in[2]], Int64[shape_out[1], shape_out[2]]
        h = Ref{Ptr{Cvoid}}(C_NULL)
        GC.@preserve shp_in shp_out check(ccall((:pxl_generic_plan_create, libpixell_hip), Cint,
            (Ref{CarWCS}, Cint, Ptr{Int64}, Ref{CarWCS}, Cint, Ptr{Int64}, Ptr{Cvoid}, Ptr{Ptr{Cvoid}}),
            CarWCS(wcs_in), projcode(wcs_in), shp_in, CarWCS(wcs_out), projcode(wcs_out), shp_out, NULLSTREAM, h))
        p = new(h[], (shape_in[1], shape_in[2]), (shape_out[1], shape_out[2]), wcs_out)
        finalizer(x -> ccall((:pxl_generic_plan_destroy, libpixell_hip), Cint, (Ptr{Cvoid},), x.handle), p)
    end
end
function generic_plan_tiles(plan::GenericReprojectPlan)
    ex, tot = Ref{Int64}(0), Ref{Int64}(0)
    check(ccall((:pxl_generic_plan_tiles, libpixell_hip), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}), plan.handle, ex, tot))
    return ex[], tot[]
end
function reproject!(dst::HIPArray{Float64}, plan::GenericReprojectPlan, src::HIPArray{Float64})
    nc = ndims(src) == 3 ? size(src, 3) : 1
    (size(src, 1), size(src, 2)) == plan.shape_in || error("reproject!: source is $(size(src)), plan expects $(plan.shape_in)")
    length(dst) == nc * prod(plan.shape_out) || error("reproject!: destination has $(length(dst)) elements")
    GC.@preserve src dst check(ccall((:pxl_generic_plan_execute, libpixell_hip), Cint,
        (Ptr{Cvoid}, Int64, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cvoid}), plan.handle, nc, src.ptr, dst.ptr, NULLSTREAM))
    return dst
end

# ---- a kept reprojection plan (coordinate tables live on the device): create once, execute per map.
#      Float64 and Float32 storage; windows (row0, nrows) describe declination strips of a sharded map.
mutable struct ReprojectPlan
    handle::Ptr{Cvoid}
    shape_in::NTuple{3,Int}
    shape_out::NTuple{2,Int}
    function ReprojectPlan(shape_in, wcs_in::AbstractCARWCS, shape_out, wcs_out::AbstractCARWCS;
                           src_rows=(0, shape_in[2]), dst_rows=(0, shape_out[2]))
        nc = length(shape_in) > 2 ? shape_in[3] : 1
        shp_in, shp_out = Int64[shape_in[1], shape_in[2], nc], Int64[shape_out[1], shape_out[2]]
        h = Ref{Ptr{Cvoid}}(C_NULL)
        GC.@preserve shp_in shp_out check(ccall((:pxl_reproject_plan_create, libpixell_hip), Cint,
            (Ref{CarWCS}, Ptr{Int64}, Int64, Int64, Ref{CarWCS}, Ptr{Int64}, Int64, Int64, Ptr{Ptr{Cvoid}}),
            CarWCS(wcs_in), shp_in, src_rows[1], src_rows[2], CarWCS(wcs_out), shp_out, dst_rows[1], dst_rows[2], h))
        p = new(h[], (shape_in[1], shape_in[2], nc), (shape_out[1], shape_out[2]))
        finalizer(x -> ccall((:pxl_reproject_plan_destroy, libpixell_hip), Cint, (Ptr{Cvoid},), x.handle), p)
    end
end

function reproject!(dst::HIPArray{Float64}, plan::ReprojectPlan, src::HIPArray{Float64})
    GC.@preserve src dst check(ccall((:pxl_reproject_execute, libpixell_hip), Cint,
        (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cvoid}), plan.handle, src.ptr, dst.ptr, NULLSTREAM))
    return dst
end
function reproject!(dst::HIPArray{Float32}, plan::ReprojectPlan, src::HIPArray{Float32})
    GC.@preserve src dst check(ccall((:pxl_reproject_execute_f32, libpixell_hip), Cint,
        (Ptr{Cvoid}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cvoid}), plan.handle, src.ptr, dst.ptr, NULLSTREAM))
    return dst
end

# ---- scattered bilinear sample at a 2xN device batch of (ra, dec): one value per point and component
function sample_bilinear(m::Enmap{Float64,N,<:HIPArray,<:AbstractCARWCS}, sky::DevCoords) where {N}
    nc = N == 3 ? size(m, 3) : 1
    n = size(sky, 2)
    out = HIPArray{Float64}(undef, n, nc)                    # (n, nc) column-major, as the header says
    shp = Int64[size(m, 1), size(m, 2), nc]
    src = parent(m)
    GC.@preserve src sky out shp check(ccall((:pxl_sample_car_bilinear_f64, libpixell_hip), Cint,
        (Ref{CarWCS}, Ptr{Int64}, Ptr{Cdouble}, Int64, Int64, Int64, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cvoid}),
        CarWCS(getwcs(m)), shp, src.ptr, 0, size(m, 2), n, sky.ptr, out.ptr, NULLSTREAM))
    return out
end

# ---- the same sample through a row-pair copy of the map (8/3 of its footprint, one random 64-byte sector per point
#      instead of 2.25: 1.75x faster on a 0.5-arcmin map).  Build once per map, sample any number of batches.
struct SamplePairs
    data::HIPArray{Float64,1}
    wcs::CarWCS
    shape::Vector{Int64}         # (nx, ny, nc)
end
function SamplePairs(m::Enmap{Float64,N,<:HIPArray,<:AbstractCARWCS}) where {N}
    nc = N == 3 ? size(m, 3) : 1
    shp = Int64[size(m, 1), size(m, 2), nc]
    nel = ccall((:pxl_sample_pairs_elems, libpixell_hip), Int64, (Ptr{Int64}, Int64), shp, size(m, 2))
    nel < 0 && check(Cint(-22))                                   # raises with pxl_last_error()
    data = HIPArray{Float64}(undef, nel)                       # device allocations are 256-byte aligned
    src = parent(m)
    GC.@preserve src data shp check(ccall((:pxl_sample_build_pairs_f64, libpixell_hip), Cint,
        (Ptr{Int64}, Ptr{Cdouble}, Int64, Ptr{Cdouble}, Ptr{Cvoid}), shp, src.ptr, size(m, 2), data.ptr, NULLSTREAM))
    return SamplePairs(data, CarWCS(getwcs(m)), shp)
end
function sample_bilinear(p::SamplePairs, sky::DevCoords)
    n = size(sky, 2)
    out = HIPArray{Float64}(undef, n, Int(p.shape[3]))
    data = p.data; shp = p.shape
    GC.@preserve data sky out shp check(ccall((:pxl_sample_car_bilinear_pairs_f64, libpixell_hip), Cint,
        (Ref{CarWCS}, Ptr{Int64}, Ptr{Cdouble}, Int64, Int64, Int64, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cvoid}),
        p.wcs, shp, data.ptr, 0, shp[2], n, sky.ptr, out.ptr, NULLSTREAM))
    return out
end

# ---- pixareamap! (enmap_ops.jl:124-138) and unwind! / rewind! (enmap_ops.jl:15-32) on device arrays
function Pixell.pixareamap!(pixareas::Enmap{Float64,2,<:HIPArray,<:AbstractCARWCS})
    shp = Int64[size(pixareas, 1), size(pixareas, 2)]
    a = parent(pixareas)
    GC.@preserve a shp check(ccall((:pxl_pixareamap_car_f64, libpixell_hip), Cint,
        (Ref{CarWCS}, Ptr{Int64}, Int64, Int64, Ptr{Cdouble}, Ptr{Cvoid}),
        CarWCS(getwcs(pixareas)), shp, 0, size(pixareas, 2), a.ptr, NULLSTREAM))
    return pixareas
end

function Pixell.rewind!(angles::HIPArray{Float64}; period=2π, ref_angle=0.0)
    GC.@preserve angles check(ccall((:pxl_rewind_f64, libpixell_hip), Cint,
        (Ptr{Cdouble}, Int64, Cdouble, Cdouble, Ptr{Cvoid}), angles.ptr, length(angles), period, ref_angle, NULLSTREAM))
    return angles
end

# the non-mutating forms (enmap_ops.jl:10-13, 21-24) copy on the device and work in place on the copy
Pixell.rewind(angles::HIPArray{Float64}; period=2π, ref_angle=0.0) = rewind!(copy(angles); period=period, ref_angle=ref_angle)
Pixell.unwind(angles::HIPArray{Float64,N}; dims=N, period=2π, ref_angle=0.0) where {N} =
    unwind!(copy(angles); dims=dims, period=period, ref_angle=ref_angle)

# dims = 2 on a 2xN batch (what pix2sky! uses, car_proj.jl:111) or a plain vector
function Pixell.unwind!(angles::HIPArray{Float64,N}; dims=N, period=2π, ref_angle=0.0) where {N}
    @assert (N == 1) || (N == 2 && size(angles, 1) == 2 && dims == 2) "device unwind! handles vectors and 2xN batches along N"
    nrow = N == 1 ? 1 : 2
    GC.@preserve angles check(ccall((:pxl_unwind_f64, libpixell_hip), Cint,
        (Ptr{Cdouble}, Int64, Cint, Cdouble, Cdouble, Ptr{Cvoid}),
        angles.ptr, N == 1 ? length(angles) : size(angles, 2), nrow, period, ref_angle, NULLSTREAM))
    return angles
end

# ---- read_map / write_map with the image in HBM (enmap.jl:198-237).  The header goes through FITSIO / WCS exactly as in
#      the reference (same WCS conversion, same POLCCONV rule incl. its linear-indexing slip that also negates I); the data
#      block -- big-endian BITPIX -64 or -32, NAXIS1 = RA fastest, i.e. already Julia's layout -- is read raw, copied to the
#      device and byte-swapped there (pxl_fits_decode_f64).  Primary HDU, whole image (`sel` stays with the CPU method).
function fits_data_offset(path::String)
    open(path, "r") do io
        block = Vector{UInt8}(undef, 2880)
        off = 0
        while true
            read!(io, block)
            off += 2880
            for c in 0:35
                String(block[80c+1:80c+8]) == "END     " && return off
            end
        end
    end
end
function Pixell.read_map(path::String, ::Type{HIPArray}; verbose=true)
    f = FITS(path, "r")
    header = read_header(f[1])
    close(f)
    bitpix = header["BITPIX"]
    @assert bitpix == -64 || bitpix == -32 "device read_map handles BITPIX -64 / -32 images"
    dims = ntuple(i -> Int(header["NAXIS$i"]), header["NAXIS"])
    n = prod(dims)
    raw = Vector{UInt8}(undef, n * (abs(bitpix) ÷ 8))
    open(path, "r") do io
        seek(io, fits_data_offset(path))
        read!(io, raw)
    end
    if "STOKES" in header.values && get(header, "POLCCONV", "COSMO") == "IAU"
        # resolve_polcconv! (enmap.jl:178-195) as the reference executes it: `signs[signs_size] .= -1` is a linear index,
        # so planes 1 AND 3 of the STOKES axis change sign.  Done on the raw big-endian bytes before the upload: the sign
        # bit of an IEEE number is the top bit of its first byte.
        verbose && println("convert to IAU: flip U (and, like the reference, I)")
        esz = abs(bitpix) ÷ 8
        plane = dims[1] * dims[2]
        for c in (1, 3)
            c * plane <= n || continue
            @inbounds for k in (c - 1) * plane : c * plane - 1
                raw[esz * k + 1] ⊻= 0x80
            end
        end
    end
    draw = HIPArray(raw)
    data = HIPArray{Float64}(undef, dims...)
    GC.@preserve draw data check(ccall((:pxl_fits_decode_f64, libpixell_hip), Cint,
        (Ptr{Cvoid}, Ptr{Cdouble}, Int64, Cint, Ptr{Cvoid}), draw.ptr, data.ptr, n, bitpix, NULLSTREAM))
    header_str = join([@sprintf("%-80s", l) for l in split(string(header), "\n")])
    wcs0 = WCS.from_header(header_str)[1]
    @assert wcs0.ctype[1] == "RA---CAR" && wcs0.ctype[2] == "DEC--CAR"
    wcs = convert(CarClenshawCurtis{Float64}, wcs0)
    return Enmap(data, wcs)
end

function Pixell.write_map(fname::String, emap::Enmap{Float64,N,<:HIPArray}) where {N}
    data = parent(emap)
    n = length(data)
    draw = HIPArray{UInt8}(undef, 8n)
    GC.@preserve data draw check(ccall((:pxl_fits_encode_f64, libpixell_hip), Cint,
        (Ptr{Cdouble}, Ptr{Cvoid}, Int64, Ptr{Cvoid}), data.ptr, draw.ptr, n, NULLSTREAM))
    raw = Array(draw)
    cards = String[@sprintf("%-8s= %20s", "SIMPLE", "T"), @sprintf("%-8s= %20d", "BITPIX", -64), @sprintf("%-8s= %20d", "NAXIS", N)]
    for i in 1:N
        push!(cards, @sprintf("%-8s= %20d", "NAXIS$i", size(data, i)))
    end
    push!(cards, @sprintf("%-8s= %20s", "EXTEND", "T"))
    header = WCS.to_header(Base.convert(WCSTransform, getwcs(emap)))       # the reference's own card source (enmap.jl:229)
    append!(cards, [header[1+(i-1)*80:i*80] for i in 1:round(Int, length(header) / 80)])
    push!(cards, "END")
    open(fname, "w") do io
        for c in cards
            write(io, rpad(c, 80)[1:80])
        end
        write(io, " "^(mod(-80 * length(cards), 2880)))
        write(io, raw)
        write(io, zeros(UInt8, mod(-length(raw), 2880)))
    end
    return nothing
end

# ---- one step of the dec-strip sharded operator on this rank (one process per GPU): the library exchanges the halo
#      rows over the host's RCCL communicator (`comm::Ptr{Cvoid}` = ncclComm_t) and orders interior / boundary rows
struct HaloXfer                    # == struct pxl_halo_xfer
    peer::Int32
    reserved::Int32
    row0::Int64                    # 0-based absolute source row
    nrows::Int64
end
function sharded_step!(dst::HIPArray{Float64}, plan::ReprojectPlan, src::HIPArray{Float64}, own_rows::Tuple{Int,Int},
                       sends::Vector{HaloXfer}, recvs::Vector{HaloXfer}, comm::Ptr{Cvoid})
    GC.@preserve src dst sends recvs check(ccall((:pxl_reproject_sharded_step_f64, libpixell_hip), Cint,
        (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Int64, Int64, Ptr{HaloXfer}, Cint, Ptr{HaloXfer}, Cint, Ptr{Cvoid}, Ptr{Cvoid}),
        plan.handle, src.ptr, dst.ptr, own_rows[1], own_rows[2], sends, length(sends), recvs, length(recvs), comm, NULLSTREAM))
    return dst
end

# ---- the RCCL communicator of a Julia host (no RCCL.jl needed): rank 0 makes the 128-byte id and hands it to the other
#      ranks by any means (a file, MPI.jl, Sockets); every rank then joins with its HIP device current.
struct PxlComm
    handle::Ptr{Cvoid}
end
function comm_unique_id()
    id = Vector{UInt8}(undef, 128)                                    # PXL_COMM_ID_BYTES
    GC.@preserve id check(ccall((:pxl_comm_unique_id, libpixell_hip), Cint, (Ptr{UInt8},), id))
    return id
end
function comm_init_rank(id::Vector{UInt8}, rank::Integer, nranks::Integer)
    length(id) == 128 || error("the communicator id is 128 bytes")
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve id check(ccall((:pxl_comm_init_rank, libpixell_hip), Cint, (Ptr{UInt8}, Cint, Cint, Ptr{Ptr{Cvoid}}),
                                id, rank, nranks, h))
    return PxlComm(h[])
end
sharded_step!(dst, plan, src, own_rows, sends, recvs, comm::PxlComm) = sharded_step!(dst, plan, src, own_rows, sends, recvs, comm.handle)
comm_destroy(c::PxlComm) = check(ccall((:pxl_comm_destroy, libpixell_hip), Cint, (Ptr{Cvoid},), c.handle))
comm_backend() = unsafe_string(ccall((:pxl_comm_backend, libpixell_hip), Cstring, ()))

# ---- class-aware placement of a (source, destination) pair (DESIGN.md section 4.7; pixell.jl_amd/placement.py is the Python twin).
# The memory of an allocation falls into three classes; a kernel with several far-apart write fronts (the reprojection: one per
# XCD) stores at 5.8-6.0 TB/s into one class and at 6.8-7.1 TB/s when its destination straddles two.  `mem_probe_pair` times the
# 8-front store probe on two windows (it overwrites them with zeros): slow = same class.
function mem_probe_pair(a::Ptr, b::Ptr; window::Integer=1 << 30, reps::Integer=3)
    us = Ref{Cfloat}(0)
    check(ccall((:pxl_mem_probe_pair, libpixell_hip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Cint, Ptr{Cfloat}, Ptr{Cvoid}),
                Ptr{Cvoid}(a), Ptr{Cvoid}(b), window, reps, us, NULLSTREAM))
    return us[]
end

# The same rule natively (pxl_mem_pair_alloc): one ccall, tested from the Python side on the device (tests/test_gpu_placement.py)
mutable struct MemPair                 # == struct pxl_mem_pair
    src::Ptr{Cvoid}
    dst::Ptr{Cvoid}
    arena::Ptr{Cvoid}
    src_alloc::Ptr{Cvoid}
    arena_bytes::UInt64
    src_offset::UInt64
    dst_offset::UInt64
    classes::Int32
    dst_two_classes::Int32
    src_own_class::Int32
    probes::Int32
    separate_tried::Int32
    reserved_::Int32
    MemPair() = new(C_NULL, C_NULL, C_NULL, C_NULL, 0, 0, 0, 0, 0, 0, 0, 0, 0)
end
# (src, dst, pair): device arrays over the library's allocation; `pair` owns the memory (freed by its finalizer, which the two
# arrays keep from running by holding a reference to it)
function place_pair_native(::Type{T}, src_dims::NTuple{N,Int}, dst_dims::NTuple{M,Int}; headroom::Integer=144 << 30) where {T,N,M}
    pair = MemPair()
    # pointer_from_objref gives a raw address the GC knows nothing about: the object must be rooted across the call
    GC.@preserve pair begin
        check(ccall((:pxl_mem_pair_alloc, libpixell_hip), Cint, (UInt64, UInt64, UInt64, Ptr{Cvoid}, Ptr{Cvoid}),
                    prod(src_dims) * sizeof(T), prod(dst_dims) * sizeof(T), headroom, pointer_from_objref(pair), NULLSTREAM))
    end
    # (a finalizer's argument is rooted while it runs)
    finalizer(p -> GC.@preserve(p, ccall((:pxl_mem_pair_free, libpixell_hip), Cint, (Ptr{Cvoid},), pointer_from_objref(p))), pair)
    return HIPArray{T,N}(Ptr{T}(pair.src), src_dims, pair), HIPArray{T,M}(Ptr{T}(pair.dst), dst_dims, pair), pair
end

# class label of every `step`-spaced 1 GiB window of `arena` (labels 1, 2, 3 in order of appearance)
function map_classes(arena::HIPArray{UInt8,1}; step::Integer=2 << 30, window::Integer=1 << 30)
    thr = 2.0 * window / 6.25e6                       # microseconds: above = the two windows share a class (~383 vs ~305 us per GiB)
    offs = collect(0:step:(length(arena) - window))
    labels = zeros(Int, length(offs)); refs = Int[]    # refs[c] = offset of the first window seen of class c
    for (k, off) in enumerate(offs)
        for (c, r) in enumerate(refs)
            if off == r || mem_probe_pair(arena.ptr + off, arena.ptr + r; window=window) > thr
                labels[k] = c; break
            end
        end
        if labels[k] == 0
            push!(refs, off); labels[k] = length(refs)
        end
    end
    return offs, labels
end

# (src, dst): views of ONE allocation of (pair + headroom) bytes, the destination centred on a boundary between two classes when
# the allocation has one with room on both sides, the source in windows of a class the destination does not touch when there
# are any; a plain layout (source first, destination above it) otherwise.  Both are zero-filled by the probe or by hipMemset.
function place_pair(::Type{T}, src_dims::NTuple{N,Int}, dst_dims::NTuple{M,Int}; headroom::Integer=144 << 30, step::Integer=2 << 30) where {T,N,M}
    al = 2 << 20
    bs, bd = cld(prod(src_dims) * sizeof(T), al) * al, cld(prod(dst_dims) * sizeof(T), al) * al
    arena = HIPArray{UInt8,1}(undef, (bs + bd + headroom,))
    offs, labels = map_classes(arena; step=step)
    total = length(arena)
    src_off, dst_off = 0, bs
    for k in 2:length(offs)
        labels[k] == labels[k - 1] && continue
        cand = (offs[k] - bd ÷ 2) ÷ al * al            # the boundary lies between windows k-1 and k
        if cand >= 0 && cand + bd <= total
            dst_off = cand
            touched = Set(labels[j] for j in eachindex(offs) if offs[j] + (1 << 30) > dst_off && offs[j] < dst_off + bd)
            clear(o) = all(!(labels[j] in touched) for j in eachindex(offs) if offs[j] + (1 << 30) > o && offs[j] < o + bs)
            free = [o for o in offs if (o + bs <= dst_off || o >= dst_off + bd) && o + bs <= total && clear(o)]
            src_off = !isempty(free) ? first(free) : (dst_off >= bs ? 0 : dst_off + bd)
            src_off + bs <= total || continue
            break
        end
    end
    ccall((:hipMemset, libhip), Cint, (Ptr{Cvoid}, Cint, Csize_t), arena.ptr + src_off, 0, bs)
    src = HIPArray{T,N}(Ptr{T}(arena.ptr + src_off), src_dims, arena)
    dst = HIPArray{T,M}(Ptr{T}(arena.ptr + dst_off), dst_dims, arena)
    return src, dst
end

export mem_probe_pair, map_classes, place_pair, place_pair_native, MemPair, MemPlacedInfo, ALLOC_POLICY
export HIPArray, posmap_device, reproject, reproject_generic, reproject!, ReprojectPlan, GenericReprojectPlan, generic_plan_tiles, sample_bilinear, SamplePairs, HaloXfer, sharded_step!
export PxlComm, comm_unique_id, comm_init_rank, comm_destroy, comm_backend
end # module
