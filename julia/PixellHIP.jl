# PixellHIP.jl -- the reference-side binding: new methods of Pixell's own generic functions for maps whose
# storage lives in MI355X HBM, each a thin `ccall` into libpixell_hip.so (include/pixell_hip.h).
#
# NOT EXERCISED IN THIS REPOSITORY'S CI: Julia is not installed in the build image (see DESIGN.md 1); the
# executable twin of this file is the Python host package pixell.jl_amd/.  It follows the reference's own FFI
# style (`ccall((:sym, lib), Cint, (...), ...)` + `GC.@preserve`, src/transforms.jl:185-194) but checks the
# returned code.  No CUDA.jl / AMDGPU.jl: device memory is a minimal array type over hipMalloc.
module PixellHIP

using Pixell
import Pixell: Enmap, AbstractCARWCS, CarClenshawCurtis, getwcs, pix2sky, pix2sky!, sky2pix, sky2pix!, posmap,
               pixareamap!, rewind!, unwind!

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

function check(rc::Cint)
    rc == 0 && return nothing
    buf = Vector{UInt8}(undef, 512)
    ccall((:pxl_last_error, libpixell_hip), Csize_t, (Ptr{UInt8}, Csize_t), buf, 512)
    error("libpixell_hip ($rc): " * unsafe_string(pointer(buf)))     # the reference raises exceptions too
end

# ---- minimal device array (column-major, like Array): the `AA` slot of Enmap{T,N,AA,W} (enmap.jl:10)
mutable struct HIPArray{T,N} <: AbstractArray{T,N}
    ptr::Ptr{T}
    dims::NTuple{N,Int}
    function HIPArray{T,N}(::UndefInitializer, dims::NTuple{N,Int}) where {T,N}
        p = Ref{Ptr{Cvoid}}()
        rc = ccall((:hipMalloc, libhip), Cint, (Ptr{Ptr{Cvoid}}, Csize_t), p, prod(dims) * sizeof(T))
        rc == 0 || error("hipMalloc failed ($rc)")
        a = new{T,N}(Ptr{T}(p[]), dims)
        finalizer(x -> ccall((:hipFree, libhip), Cint, (Ptr{Cvoid},), x.ptr), a)
    end
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

const DevCoords = HIPArray{Float64,2}                      # 2xN, interleaved pairs (car_proj.jl:102-107)
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

# dims = 2 on a 2xN batch (what pix2sky! uses, car_proj.jl:111) or a plain vector
function Pixell.unwind!(angles::HIPArray{Float64,N}; dims=N, period=2π, ref_angle=0.0) where {N}
    @assert (N == 1) || (N == 2 && size(angles, 1) == 2 && dims == 2) "device unwind! handles vectors and 2xN batches along N"
    nrow = N == 1 ? 1 : 2
    GC.@preserve angles check(ccall((:pxl_unwind_f64, libpixell_hip), Cint,
        (Ptr{Cdouble}, Int64, Cint, Cdouble, Cdouble, Ptr{Cvoid}),
        angles.ptr, N == 1 ? length(angles) : size(angles, 2), nrow, period, ref_angle, NULLSTREAM))
    return angles
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

export HIPArray, posmap_device, reproject, reproject!, ReprojectPlan, sample_bilinear, SamplePairs, HaloXfer, sharded_step!
export PxlComm, comm_unique_id, comm_init_rank, comm_destroy, comm_backend
end # module
