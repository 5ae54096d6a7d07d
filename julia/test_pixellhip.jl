# test_pixellhip.jl -- the reference's own known answers, through the ccall binding (julia/PixellHIP.jl).
# NOT RUN IN THIS REPOSITORY'S CI (no Julia in the build image); kept ready for a box that has
# Julia >= 1.6, Pixell.jl and an MI355X:
#     PIXELL_HIP_LIB=/path/to/libpixell_hip.so julia --project test_pixellhip.jl
# The literals are those of /root/reference/test/test_geometry.jl (cited per test).
using Test
using Pixell
include("PixellHIP.jl")
using .PixellHIP

@testset "device pix2sky / sky2pix vs the reference CPU path (bit for bit)" begin
    shape, wcs = fullsky_geometry(deg2rad(1))                         # test_geometry.jl:50
    pix = 400 .* rand(2, 100_000) .- 20
    dpix = HIPArray(pix)
    # safe=false: affine only (test_geometry.jl:67-72 compares this path with wcslib)
    @test Array(pix2sky(shape, wcs, dpix; safe=false)) == pix2sky(shape, wcs, pix; safe=false)
    # safe=true: rewind + DSP.unwrap along the point axis (car_proj.jl:110-112)
    @test Array(pix2sky(shape, wcs, dpix; safe=true)) == pix2sky(shape, wcs, pix; safe=true)
    sky = pix2sky(shape, wcs, pix; safe=false)
    dsky = HIPArray(sky)
    for safe in (true, false)
        @test Array(sky2pix(shape, wcs, dsky; safe=safe)) == sky2pix(shape, wcs, sky; safe=safe)
    end
end

@testset "known answers (test_geometry.jl:52-64)" begin
    shape, wcs = fullsky_geometry(deg2rad(1))
    pix = [2.0 11.0 41.0; 2.0 -12.0 -29.0]
    sky = Array(pix2sky(shape, wcs, HIPArray(pix); safe=true))
    @test sky[:, 1] ≈ [3.12413936, -1.55334303]
    @test sky[:, 2] ≈ [2.96705973, -1.79768913]
    @test sky[:, 3] ≈ [2.44346095, -2.0943951]
    back = Array(sky2pix(shape, wcs, HIPArray(sky .+ [12π, 16π]); safe=true))      # 2*pi*k invariance, :63-64
    @test back[:, 1] ≈ [2.0, 2.0]
end

@testset "posmap (enmap_ops.jl:190-203) and range invariants (test_geometry.jl:207-223)" begin
    shape, wcs = fullsky_geometry(deg2rad(1))
    ra, dec = PixellHIP.posmap_device(shape, wcs)
    ra0, dec0 = posmap(shape, wcs)
    @test Array(parent(ra)) == parent(ra0)
    @test Array(parent(dec)) == parent(dec0)
    @test all(-π .≤ Array(parent(ra)) .≤ π) && all(-π / 2 .≤ Array(parent(dec)) .≤ π / 2)
end

@testset "reproject: identity, partition of unity, half-pixel shift across the RA seam" begin
    shape, wcs = fullsky_geometry(2π / 256)
    m = Enmap(HIPArray(randn(shape...)), wcs)
    same = PixellHIP.reproject(m, shape, wcs)
    @test maximum(abs.(Array(parent(same)) .- Array(parent(m)))) < 1e-12
    ones_map = Enmap(HIPArray(ones(shape...)), wcs)
    shape2, wcs2 = fullsky_geometry(2π / 512)
    @test maximum(abs.(Array(parent(PixellHIP.reproject(ones_map, shape2, wcs2))) .- 1)) < 1e-13
    shifted = CarClenshawCurtis{Float64}(wcs.cdelt, wcs.crpix .- (0.5, 0.0), wcs.crval, wcs.unit)
    out = Array(parent(PixellHIP.reproject(m, shape, shifted)))
    src = Array(parent(m))
    @test maximum(abs.(out .- 0.5 .* (src .+ circshift(src, (-1, 0))))) < 1e-12
end

@testset "kept plan, Float32 storage, sampler, pixareamap!, unwind!" begin
    shape, wcs = fullsky_geometry(2π / 256)
    shape2, wcs2 = fullsky_geometry(2π / 512)
    plan = PixellHIP.ReprojectPlan(shape, wcs, shape2, wcs2)
    src = randn(shape...)
    out64 = Array(PixellHIP.reproject!(HIPArray{Float64}(undef, shape2...), plan, HIPArray(src)))
    @test out64 == Array(parent(PixellHIP.reproject(Enmap(HIPArray(src), wcs), shape2, wcs2)))
    out32 = Array(PixellHIP.reproject!(HIPArray{Float32}(undef, shape2...), plan, HIPArray(Float32.(src))))
    @test maximum(abs.(out32 .- Float32.(out64))) < 1f-5
    # sampling at the output pixel centres reproduces the reprojection (to rounding: two sky2pix roundings)
    m = Enmap(HIPArray(src), wcs)
    ra, dec = posmap(shape2, wcs2)
    sky = HIPArray(permutedims(hcat(vec(parent(ra)), vec(parent(dec)))))
    @test maximum(abs.(vec(Array(PixellHIP.sample_bilinear(m, sky))) .- vec(out64))) < 1e-9
    # pixareamap! against the reference's CPU method (test_geometry.jl:287-316)
    a_dev = Enmap(HIPArray{Float64}(undef, shape...), wcs)
    @test maximum(abs.(Array(parent(pixareamap!(a_dev))) .- parent(pixareamap(Enmap(zeros(shape...), wcs))))) < 100eps()
    # unwind! along the point axis of a 2xN batch (enmap_ops.jl:26-32), bit for bit
    ang = cumsum(1.5 .* randn(2, 50_000), dims=2)
    @test Array(unwind!(HIPArray(copy(ang)); dims=2)) == unwind!(copy(ang); dims=2)
end

@testset "two-vector forms (car_proj.jl:141-152, 235-252) against the reference's CPU methods, bit for bit" begin
    shape, wcs = fullsky_geometry(deg2rad(1))
    ip, jp = 400 .* rand(100_000) .- 20, 200 .* rand(100_000) .- 10
    for safe in (true, false)
        ra, dec = pix2sky(shape, wcs, HIPArray(ip), HIPArray(jp); safe=safe)
        ra0, dec0 = pix2sky(shape, wcs, ip, jp; safe=safe)
        @test Array(ra) == ra0 && Array(dec) == dec0
        x, y = sky2pix(shape, wcs, HIPArray(ra0), HIPArray(dec0); safe=safe)       # the ::AV method: reciprocal-period form
        x0, y0 = sky2pix(shape, wcs, ra0, dec0; safe=safe)
        @test Array(x) == x0 && Array(y) == y0
    end
    # docstring literals: (30, 80) -> (151 deg, -11 deg); (30 deg, 80 deg) -> (151, 171)   car_proj.jl:137-138, 216-217
    ra, dec = pix2sky(shape, wcs, HIPArray([30.0]), HIPArray([80.0]))
    @test rad2deg.([Array(ra)[1], Array(dec)[1]]) ≈ [151.0, -11.0]
    x, y = sky2pix(shape, wcs, HIPArray([deg2rad(30.0)]), HIPArray([deg2rad(80.0)]))
    @test [Array(x)[1], Array(y)[1]] ≈ [151.0, 171.0]
end

@testset "Gnomonic evaluators and posmap (tan_proj.jl:44-75; test_geometry.jl:92-119)" begin
    wcs = Pixell.Gnomonic{Float64}((-0.004166666666666667, 0.004166666666666667), (914.0, 913.0), (53.0, -28.0), π / 180)
    shape = (1827, 1825)
    ra, dec = PixellHIP.posmap_device(shape, wcs)
    ra0, dec0 = posmap(shape, wcs)
    # the reference's own bound for its fast Gnomonic path against wcslib: summed absolute difference below 1e-9
    @test sum(abs.(Array(parent(ra)) .- parent(ra0))) < 1e-9
    @test sum(abs.(Array(parent(dec)) .- parent(dec0))) < 1e-9
    i, j = 1800 .* rand(10_000), 1800 .* rand(10_000)
    a, d = pix2sky(shape, wcs, HIPArray(i), HIPArray(j))
    a0 = [pix2sky(shape, wcs, i[k], j[k])[1] for k in eachindex(i)]
    @test maximum(abs.(Array(a) .- a0)) < 1e-13
    x, y = sky2pix(shape, wcs, a, d)
    @test maximum(abs.(Array(x) .- i)) < 1e-8 && maximum(abs.(Array(y) .- j)) < 1e-8
end

@testset "CAR <-> Gnomonic reprojection, non-mutating rewind / unwind, FITS round trip" begin
    shape, wcs = fullsky_geometry(2π / 2160)
    m = Enmap(HIPArray(ones(shape...)), wcs)
    tan = Pixell.Gnomonic{Float64}((-1 / 6, 1 / 6), (128.5, 128.5), (30.0, 10.0), π / 180)
    patch = PixellHIP.reproject(m, (256, 256), tan)                     # picked by dispatch on the Gnomonic output WCS
    @test maximum(abs.(Array(parent(patch)) .- 1)) < 1e-12
    plan = PixellHIP.GenericReprojectPlan(shape, wcs, (256, 256), tan)     # the lattice kept: same bits as the one-shot call
    kept = PixellHIP.reproject!(HIPArray{Float64}(undef, 256, 256), plan, parent(m))
    @test Array(kept) == Array(parent(patch))
    @test PixellHIP.generic_plan_tiles(plan) == (0, 2 * 8)                 # 128 x 32 tiles, none evaluated per pixel
    back = PixellHIP.reproject(patch, shape, wcs)                       # Gnomonic -> CAR: ones inside the patch, zeros outside
    @test all(x -> abs(x) < 1e-12 || abs(x - 1) < 1e-9 || 0 <= x <= 1, Array(parent(back)))
    ang = 40 .* rand(2, 10_000) .- 20
    d = HIPArray(ang)
    @test Array(rewind(d)) == rewind(ang) && Array(d) == ang            # the input is left alone (enmap_ops.jl:10-13)
    @test Array(unwind(d; dims=2)) == unwind(ang; dims=2) && Array(d) == ang
    path = joinpath(mktempdir(), "roundtrip.fits")
    src = randn(shape...)
    write_map(path, Enmap(HIPArray(src), wcs))                          # big-endian encode on the device
    @test parent(read_map(path)) == src                                 # the reference's CPU reader sees the same map
    dm = read_map(path, HIPArray)
    @test Array(parent(dm)) == src && getwcs(dm) == getwcs(read_map(path))
    ref = read_map(joinpath(@__DIR__, "..", "tests", "golden", "test.fits"))   # the reference's own fixture (test/data/test.fits)
    @test Array(parent(read_map(joinpath(@__DIR__, "..", "tests", "golden", "test.fits"), HIPArray))) == parent(ref)
end

@testset "class-aware placement of a reprojection pair" begin
    src, dst = PixellHIP.place_pair(Float64, (21600, 10801), (43200, 21601); headroom=64 << 30)
    @test size(src) == (21600, 10801) && size(dst) == (43200, 21601) && src.parent === dst.parent
    lo, hi = minmax(UInt(src.ptr), UInt(dst.ptr))                       # the views do not overlap and lie inside the allocation
    @test lo + (lo == UInt(src.ptr) ? sizeof(Float64) * prod(size(src)) : sizeof(Float64) * prod(size(dst))) <= hi
    @test all(==(0.0), Array(src)[1:1000])                              # the source is zero-filled
    offs, labels = PixellHIP.map_classes(src.parent)
    @test 1 <= maximum(labels) <= 3                                     # the three memory classes of the part (DESIGN.md section 4.7)
    plan = PixellHIP.ReprojectPlan(fullsky_geometry(deg2rad(1 / 60))..., fullsky_geometry(deg2rad(0.5 / 60))...)
    PixellHIP.reproject!(dst, plan, src)
    @test all(==(0.0), Array(dst)[1:1000])
end

@testset "the library's allocation policy behind HIPArray(undef, ...) / similar" begin
    # enmap.jl:60-62: `similar` keeps the array type, so every fresh device map comes from HIPArray{T,N}(undef, dims), i.e. from
    # pxl_mem_alloc_placed (include/pixell_hip.h): map-sized buffers across two memory classes, no head-room kept; small ones plain
    @test PixellHIP.ALLOC_POLICY[] === :class_aware
    big = HIPArray{Float64}(undef, 43200, 21601)                        # 7.5 GB: goes through the class search
    @test size(big) == (43200, 21601) && UInt(big.ptr) % 256 == 0
    small = similar(big, Float64, (128, 64))                            # below 3 GiB: a plain allocation through the same entry
    @test size(small) == (128, 64)
    m = Enmap(HIPArray(ones(360, 181)), fullsky_geometry(deg2rad(1))[2])
    @test similar(m) isa Enmap && parent(similar(m)) isa HIPArray       # the reference's own `similar(::Enmap)` lands here
    PixellHIP.ALLOC_POLICY[] = :plain
    try
        plain = HIPArray{Float64}(undef, 4096, 4096)
        @test size(plain) == (4096, 4096)
    finally
        PixellHIP.ALLOC_POLICY[] = :class_aware
    end
end
