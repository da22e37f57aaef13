# reference_golden.jl -- ONE command that pins the oracle to the reference, for whoever has Julia 0.6 + Celeste.jl:
#
#     julia tools/reference_golden.jl [fixture ...]        (default: every tests/golden/raw/*.txt)
#
# For every fixture exported by tests/golden/export_raw.py it rebuilds the inputs with the REFERENCE's own constructors
# (Model.Image, ConstantPSFMap, CatalogEntry, Model.get_sky_patches, Model.find_neighbors, ElboArgs), evaluates
# DeterministicVI.elbo(ea, vp) for every source as the single active source with its neighbours value-only -- the call of
# ParallelRun.process_source (src/ParallelRun.jl:468-488) -- and writes value, gradient and Hessian to
# tests/golden/ref/<fixture>.txt.  tests/test_reference_outputs.py then holds the C oracle (and, with -m gpu, the HIP
# engine) to those numbers at the 1e-8 of BASELINE.json: with such a file present, parity is pinned by the reference itself.
#
# NOT RUN by the builder (no Julia in the build image or on the GPU box): written against the reference sources
# (src/model/image_model.jl:6-46, src/model/psf_model.jl:17-29, 87-95, src/model/imaged_sources.jl:165-244,
# src/deterministic_vi/elbo_args.jl:165-211, src/deterministic_vi/elbo_objective.jl:482-492, test/SampleData.jl:28-32).
using Celeste
using Celeste: Model, DeterministicVI
import WCS
using StaticArrays

const ROOT = normpath(joinpath(dirname(@__FILE__), ".."))
const RAW = joinpath(ROOT, "tests", "golden", "raw")
const OUT = joinpath(ROOT, "tests", "golden", "ref")

# one line of the manifest: name dtype ndims dim1 [dim2 ...] byte_offset; arrays are little-endian, column-major
function read_arrays(name)
    arrays = Dict{String,Any}()
    types = Dict("int64" => Int64, "int32" => Int32, "float32" => Float32, "float64" => Float64)
    open(joinpath(RAW, name * ".bin")) do io
        for line in eachline(joinpath(RAW, name * ".txt"))
            tok = split(strip(line))
            isempty(tok) && continue
            T = types[tok[2]]
            nd = parse(Int, tok[3])
            dims = ntuple(k -> parse(Int, tok[3 + k]), nd)
            seek(io, parse(Int, tok[4 + nd]))
            arrays[tok[1]] = read(io, T, dims)
        end
    end
    arrays
end

# the world coordinate system in which world and pixel coordinates coincide (test/SampleData.jl:28-32)
const wcs_id = WCS.WCSTransform(2, cd = Float64[1 0; 0 1], ctype = ["none", "none"], crpix = Float64[1, 1],
                                crval = Float64[1, 1])

function build_images(a)
    N = Int(a["n_images"][1])
    images = Model.Image[]
    for n in 1:N
        p = a["psf_$n"]                       # K x 6: alphaBar, xiBar[1:2], tauBar[1,1], tauBar[1,2], tauBar[2,2]
        psf = [Model.PsfComponent(p[k, 1], SVector{2,Float64}(p[k, 2], p[k, 3]),
                                  SMatrix{2,2,Float64,4}(p[k, 4], p[k, 5], p[k, 5], p[k, 6])) for k in 1:size(p, 1)]
        push!(images, Model.Image(a["pixels_$n"], Int(a["band_$n"][1]), wcs_id, psf, a["sky_$n"], a["nelec_per_nmgy_$n"],
                                  Model.ConstantPSFMap(a["psf_stamp_$n"])))
    end
    images
end

function build_catalog(a)
    S = Int(a["n_sources"][1])
    [Model.CatalogEntry(vec(a["pos"][s, :]), a["is_star"][s] != 0, vec(a["star_fluxes"][s, :]), vec(a["gal_fluxes"][s, :]),
                        a["gal_shape"][s, 1], a["gal_shape"][s, 2], a["gal_shape"][s, 3], a["gal_shape"][s, 4]) for s in 1:S]
end

function run_fixture(name)
    a = read_arrays(name)
    images = build_images(a)
    catalog = build_catalog(a)
    S = length(catalog)
    vp = [vec(a["vp"][s, :]) for s in 1:S]
    patches = Model.get_sky_patches(images, catalog)
    mkpath(OUT)
    open(joinpath(OUT, name * ".txt"), "w") do io
        println(io, "# DeterministicVI.elbo(ElboArgs(images, patches[[s; neighbors], :], [1]), vp) for every source s of ", name)
        println(io, "# lines: `v s value`, `d s 44 values`, `h s 1936 values (column-major)`, `n s neighbours...`; s is 0-based")
        for s in 1:S
            nbrs = Model.find_neighbors(patches, s)
            ids = vcat([s], nbrs)
            ea = DeterministicVI.ElboArgs(images, patches[ids, :], [1])
            r = DeterministicVI.elbo(ea, vp[ids])
            @printf(io, "v %d %.17g\n", s - 1, r.v[])
            print(io, "d ", s - 1); for x in r.d[:, 1]; @printf(io, " %.17g", x); end; println(io)
            print(io, "h ", s - 1); for x in r.h[1:44, 1:44]; @printf(io, " %.17g", x); end; println(io)
            print(io, "n ", s - 1); for t in nbrs; print(io, " ", t - 1); end; println(io)
        end
    end
    println("wrote ", joinpath(OUT, name * ".txt"))
end

names = isempty(ARGS) ? [splitext(f)[1] for f in readdir(RAW) if endswith(f, ".txt")] : ARGS
for nm in names
    run_fixture(nm)
end
