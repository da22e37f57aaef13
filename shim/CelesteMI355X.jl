# CelesteMI355X.jl -- ccall shim for libceleste_mi355x.so (include/celeste_mi355x.h), Julia 0.6 syntax.
#
# Source only: Julia is not installed in the build container or on the GPU box, so this file has NOT been executed.
# The tested counterpart is the ctypes binding celeste.jl_amd/cabi.py, which issues the same call sequence
# (tests/test_gpu_*.py).  To be included next to src/deterministic_vi/elbo_objective.jl inside module DeterministicVI;
# see INTEGRATION.md for the call-site changes (ElboMaximize.jl:166, ParallelRun.jl:372-397, 468-498).

const libceleste = "libceleste_mi355x"

struct CImage                      # celeste_image_t
    H::Int32; W::Int32; band::Int32; reserved::Int32
    pixels::Ptr{Float32}; sky::Ptr{Float32}; nelec_per_nmgy::Ptr{Float32}
end
struct CPatch                      # celeste_patch_t
    off_h::Int32; off_w::Int32; H2::Int32; W2::Int32
    bitmap::Ptr{UInt8}
    wcs_jacobian::NTuple{4,Float64}; world_center::NTuple{2,Float64}; pixel_center::NTuple{2,Float64}
    psf::Ptr{Float64}; stamp::Int32; reserved::Int32
end
struct CProblem                    # celeste_problem_t
    n_images::Int32; n_sources::Int32; psf_K::Int32; n_stamps::Int32
    images::Ptr{CImage}; patches::Ptr{CPatch}; stamps::Ptr{Float64}
    nbr_offsets::Ptr{Int64}; nbr_index::Ptr{Int32}; prior::Ptr{Void}
    n_patch_entries::Int64; patch_source::Ptr{Int32}; patch_image::Ptr{Int32}   # 0 / NULL: dense patch table
end

check(st) = st == 0 || error(unsafe_string(ccall((:celeste_strerror, libceleste), Cstring, (Cint,), st)))

# the struct layouts below (COptimConfig ...) are those of ABI version 2xx (CELESTE_ABI_VERSION in the header)
const CELESTE_ABI_MAJOR = 2
function check_abi()
    v = ccall((:celeste_version, libceleste), Cint, ())
    div(v, 100) == CELESTE_ABI_MAJOR || error("libceleste_mi355x has ABI version $v, this shim was written against $(CELESTE_ABI_MAJOR)xx")
end

# ---- the images of a box, uploaded ONCE (celeste_images_create) ------------------------------------------------
# process_source builds one ElboArgs per source over the same `images` (ParallelRun.jl:468-488); every per-source
# context below is created on this handle and costs a patch-table upload, not a copy of the planes: 0.17 ms to create, 0.09 ms
# for its first call, 0.02 ms to destroy (its streams, page-locked staging and upload arena come from the library's pools and
# go back to them: celeste_ctx_destroy in the header), 67 us per elbo() after that.
mutable struct MI355XImages
    handle::Ptr{Void}
    images::Vector{Image}
end
function MI355XImages(images::Vector{Image}; device::Int = 0)
    check_abi()
    keep = Any[]
    cimgs = map(images) do img
        sky = convert(Matrix{Float32}, img.sky)          # SDSSBackground / Fill materialised: img.sky[h, w]
        push!(keep, sky)
        CImage(img.H, img.W, img.b, 0, pointer(img.pixels), pointer(sky), pointer(img.nelec_per_nmgy))
    end
    h = Ref{Ptr{Void}}(C_NULL)
    check(ccall((:celeste_images_create, libceleste), Cint, (Int32, Ptr{CImage}, Cint, Ref{Ptr{Void}}),
                length(cimgs), cimgs, device, h))       # the planes are copied: `keep` may go after the call
    imgs = MI355XImages(h[], images)
    finalizer(imgs, x -> ccall((:celeste_images_destroy, libceleste), Void, (Ptr{Void},), x.handle))
    imgs
end

mutable struct MI355XContext
    handle::Ptr{Void}
    images::MI355XImages
end

"""
Device context for one ElboArgs (ea.images, ea.patches[S x N], ea.active_sources == [1]): the local source list is
[target; neighbors] exactly as ParallelRun.process_source builds it.  `imgs` is the box's image handle.
"""
function MI355XContext(ea::ElboArgs, imgs::MI355XImages)
    keep = Any[]
    stamps = Float64[]; stamp_id = Dict{Tuple{Int,UInt64},Int32}(); patches = CPatch[]
    for s in 1:ea.S, n in 1:ea.N
        p = ea.patches[s, n]
        psf = vcat([[pc.alphaBar, pc.xiBar[1], pc.xiBar[2], pc.tauBar[1,1], pc.tauBar[1,2], pc.tauBar[2,2]]
                    for pc in p.psf]...)
        bm = convert(Matrix{UInt8}, p.active_pixel_bitmap); push!(keep, psf, bm)
        # raw psfmap stamp at the patch centre (conditioning + prefilter happen inside the library); patches that share
        # a stamp -- every patch of an image with a ConstantPSFMap -- share one entry of the table
        raw = ea.images[n].psfmap(p.pixel_center[1], p.pixel_center[2])
        id = get!(stamp_id, (n, hash(raw))) do
            append!(stamps, vec(raw)); Int32(length(stamp_id))
        end
        push!(patches, CPatch(p.bitmap_offset[1], p.bitmap_offset[2], size(bm, 1), size(bm, 2), pointer(bm),
                              tuple(p.wcs_jacobian...), tuple(p.world_center...), tuple(p.pixel_center...),
                              pointer(psf), id, 0))
    end
    # patches is laid out [s * N + n] (row s = source): transpose of Julia's column-major ea.patches
    off = Int64[0; fill(ea.S - 1, ea.S)]                 # CSR offsets: only source 1 has neighbours
    idx = Int32[1:(ea.S - 1);]
    prob = CProblem(ea.N, ea.S, ea.psf_K, length(stamp_id), C_NULL, pointer(patches), pointer(stamps),
                    pointer(off), pointer(idx), C_NULL, 0, C_NULL, C_NULL)
    h = Ref{Ptr{Void}}(C_NULL)
    # every array CProblem points into is referenced AFTER the call (the library copies what it needs during it): the
    # collector must not free `patches`, `stamps`, `off`, `idx`, the PSF vectors or the bitmaps while the ccall runs
    roots = Any[keep, patches, stamps, off, idx]
    st = ccall((:celeste_ctx_create_on, libceleste), Cint, (Ptr{Void}, Ref{CProblem}, Ref{Ptr{Void}}),
               imgs.handle, prob, h)
    length(roots) == 5 || error("unreachable")           # (keeps `roots` live until here on Julia 0.6; GC.@preserve on >= 0.7)
    check(st)
    ctx = MI355XContext(h[], imgs)
    finalizer(ctx, c -> ccall((:celeste_ctx_destroy, libceleste), Void, (Ptr{Void},), c.handle))
    ctx
end

"""Drop-in for `elbo(ea, vp, elbo_vars, bvn_bundle)` (elbo_objective.jl:482-492), Float64 only."""
function elbo(ea::ElboArgs, vp::VariationalParams{Float64}, ctx::MI355XContext,
              elbo_vars::ElboIntermediateVariables{Float64} = ElboIntermediateVariables(Float64, ea.Sa))
    @assert ea.Sa == 1 && ea.active_sources == [1]
    @assert(all(all(isfinite, vs) for vs in vp), "vp contains NaNs or Infs")
    vpm = hcat(vp...)                                   # 44 x S column-major == S rows of 44
    flags = UInt32(elbo_vars.elbo.has_gradient ? 1 : 0) | UInt32(elbo_vars.elbo.has_hessian ? 2 : 0) |
            UInt32(ea.include_kl ? 4 : 0)
    sf = elbo_vars.elbo                                 # the reference returns this scratch object too
    v = Ref{Float64}(0.0); na = Ref{Int64}(0); ni = Ref{Int64}(0)
    st = ccall((:celeste_elbo_eval, libceleste), Cint,
               (Ptr{Void}, Ptr{Float64}, Int32, UInt32, Ref{Float64}, Ptr{Float64}, Ptr{Float64},
                Ref{Int64}, Ref{Int64}),
               ctx.handle, vpm, 0, flags, v, sf.d, sf.h, na, ni)
    @assert st == 0 unsafe_string(ccall((:celeste_strerror, libceleste), Cstring, (Cint,), st))
    sf.v[] = v[]
    elbo_vars.active_pixel_counter[] += na[]; elbo_vars.inactive_pixel_counter[] += ni[]
    sf
end


# ---- no call-site change at all: the reference's own signature, dispatched on Float64 --------------------------------
# `bound_result = elbo(ea, vp, get_elbo_vars(), cfg.bvn_bundle)` (ElboMaximize.jl:166) stays as it is: a context registered
# for `ea` (register_context!, where process_source builds the ElboArgs, ParallelRun.jl:482) makes this more specific method
# -- vp::VariationalParams{Float64} against the reference's VariationalParams{T} -- evaluate on the device; an ElboArgs
# without a context, and every non-Float64 element type (the ForwardDiff.Dual calls of test/test_elbo.jl:232-237), falls
# through to the reference's method (elbo_objective.jl:482-492) via `invoke`.
# Threading contract: process_source runs inside `Threads.@threads` loops (ParallelRun.jl:285, :330), so registrations,
# releases and the look-up of the 4-argument `elbo` below race on this table -- a rehash under a concurrent `get` can miss
# (silently falling back to the CPU `elbo`) or corrupt the table.  Every access takes MI355X_CONTEXTS_LOCK (a spin lock:
# the critical sections are a dictionary operation; the device call itself runs outside the lock).  A context belongs to the
# thread that evaluates its ElboArgs: at most one call per context at a time (header: "Concurrency contract").
const MI355X_CONTEXTS = ObjectIdDict()                   # ElboArgs => MI355XContext (identity keyed; Julia 0.6: ObjectIdDict)
const MI355X_CONTEXTS_LOCK = Threads.SpinLock()
function register_context!(ea::ElboArgs, imgs::MI355XImages)
    ctx = MI355XContext(ea, imgs)                        # (the upload happens outside the lock)
    lock(MI355X_CONTEXTS_LOCK)
    try
        MI355X_CONTEXTS[ea] = ctx
    finally
        unlock(MI355X_CONTEXTS_LOCK)
    end
    ctx
end
function release_context!(ea::ElboArgs)                   # the context's finalizer frees the device tables
    lock(MI355X_CONTEXTS_LOCK)
    try
        delete!(MI355X_CONTEXTS, ea)
    finally
        unlock(MI355X_CONTEXTS_LOCK)
    end
end
function lookup_context(ea::ElboArgs)
    lock(MI355X_CONTEXTS_LOCK)
    try
        return get(MI355X_CONTEXTS, ea, nothing)
    finally
        unlock(MI355X_CONTEXTS_LOCK)
    end
end

function elbo(ea::ElboArgs, vp::VariationalParams{Float64}, elbo_vars::ElboIntermediateVariables{Float64},
              bvn_bundle::BvnBundle{Float64})
    ctx = lookup_context(ea)
    if ctx === nothing || ea.Sa != 1 || ea.active_sources != [1]
        return invoke(elbo, Tuple{ElboArgs, VariationalParams, ElboIntermediateVariables, BvnBundle}, ea, vp, elbo_vars, bvn_bundle)
    end
    elbo(ea, vp, ctx::MI355XContext, elbo_vars)          # the method above; `bvn_bundle` is not needed on the device
end
elbo(ea::ElboArgs, vp::VariationalParams{Float64}, elbo_vars::ElboIntermediateVariables{Float64}) =
    elbo(ea, vp, elbo_vars, BvnBundle{Float64}(ea.psf_K, ea.S))


# ---- whole-box entry points (one context over every catalogued source of the box; see INTEGRATION.md) ----------

"""Page-locked output buffers (celeste_host_alloc): the library DMAs results straight into them, overlapped with the
kernels of the next part of the batch.  Allocate once per box and reuse."""
mutable struct PinnedOutputs       # (mutable: a finalizer returns the page-locked blocks to the library)
    v::Vector{Float64}; d::Matrix{Float64}; h::Matrix{Float64}; counters::Matrix{Int64}; status::Vector{Int32}
    blocks::Vector{Ptr{Void}}
end
function PinnedOutputs(n::Int; packed::Bool = true)
    hs = packed ? 990 : 44 * 44                          # CELESTE_FLAG_PACKED_HESS: upper triangle by columns
    blocks = Ptr{Void}[]
    function pin(T, dims...)
        ptr = ccall((:celeste_host_alloc, libceleste), Ptr{Void}, (Csize_t,), sizeof(T) * prod(dims))
        if ptr == C_NULL                                 # (hipHostMalloc failed: give back what we have, fail loudly)
            foreach(b -> ccall((:celeste_host_free, libceleste), Void, (Ptr{Void},), b), blocks)
            error("celeste_host_alloc: out of page-locked memory")
        end
        push!(blocks, ptr)
        unsafe_wrap(Array, convert(Ptr{T}, ptr), dims)    # own = false: freed by the finalizer below, not by Julia
    end
    out = PinnedOutputs(pin(Float64, n), pin(Float64, 44, n), pin(Float64, hs, n), pin(Int64, 2, n), pin(Int32, n), blocks)
    finalizer(out, o -> foreach(b -> ccall((:celeste_host_free, libceleste), Void, (Ptr{Void},), b), o.blocks))
    out
end

"""elbo() for a conflict-free batch of targets; `targets0` are 0-based source ids.  flags: 1 gradient, 2 Hessian,
4 KL, 32 packed Hessian (element (i, j), i <= j, 0-based, at j (j + 1) / 2 + i of column t of out.h)."""
function elbo_batch!(out::PinnedOutputs, ctx::MI355XContext, vp_all::Matrix{Float64}, targets0::Vector{Int32};
                     flags::UInt32 = UInt32(7 | 32))
    st = ccall((:celeste_elbo_eval_batch, libceleste), Cint,
               (Ptr{Void}, Ptr{Float64}, Int32, Ptr{Int32}, UInt32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                Ptr{Int64}, Ptr{Int32}),
               ctx.handle, vp_all, length(targets0), targets0, flags, out.v, out.d, out.h, out.counters, out.status)
    return st      # first non-zero per-target status; out.status[t] tells which (the other targets are valid)
end

struct COptimConfig                # celeste_optim_config_t
    loc_width::Float64; loc_scale::Float64; max_iters::Int32; include_kl::Int32
    xtol_abs::Float64; ftol_rel::Float64; gtol::Float64; initial_delta::Float64; delta_hat::Float64
    tr_secular_iters::Int32; reserved::Int32     # 0: multiplier iterations to convergence; 5: Optim.jl's cap
end

"""ElboMaximize.maximize! for a conflict-free batch; vp_all (44 x S) is updated in place for the targets whose status
is 0 (a failing target keeps its column and is logged by the caller, ParallelRun.jl:389-396)."""
function maximize_batch!(ctx::MI355XContext, vp_all::Matrix{Float64}, targets0::Vector{Int32};
                         vp_frozen_neighbors = C_NULL, box_centres = C_NULL, include_kl::Bool = true,
                         tr_secular_iters::Int = 5)
    # tr_secular_iters = 5: Optim.jl's solve_tr_subproblem! stops the multiplier iteration after 5 steps -- what a run of the
    # reference does; 0 runs it to convergence (the library's own default)
    n = length(targets0)
    iterations = zeros(Int32, n); f_calls = zeros(Int32, n); max_values = zeros(n); status = zeros(Int32, n)
    cfgc = Ref(COptimConfig(1e-4, 1.0, 50, include_kl, 1e-7, 1e-6, 1e-8, 1.0, 1e9, tr_secular_iters, 0))   # ElboConfig defaults
    st = ccall((:celeste_maximize_batch, libceleste), Cint,
               (Ptr{Void}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int32, Ptr{Int32}, Ref{COptimConfig},
                Ptr{Int32}, Ptr{Int32}, Ptr{Float64}, Ptr{Int32}),
               ctx.handle, vp_all, vp_frozen_neighbors #= or C_NULL =#, box_centres #= or C_NULL =#,
               n, targets0, cfgc, iterations, f_calls, max_values, status)
    return st, iterations, f_calls, max_values, status
end

"""
ParallelRun.one_node_joint_infer's inner loop (ParallelRun.jl:302-397) in ONE call: `layers` is the schedule -- each layer a
vector of 0-based source ids no two of which are neighbours (the j-th sources of the connected components of a Cyclades
batch), repeated for every sweep; vp_all (44 x S) goes up once, stays in HBM across all layers and comes back at the end.
box_centres[k] (2 x length(layers[k])) pins the position boxes (ParallelRun.jl:96-100).  Returns the first non-zero
per-source status (those sources keep the column they had before their layer) and the per-entry outputs.
"""
function joint_infer!(ctx::MI355XContext, vp_all::Matrix{Float64}, layers::Vector{Vector{Int32}};
                      box_centres::Vector{Matrix{Float64}} = Matrix{Float64}[], include_kl::Bool = true,
                      tr_secular_iters::Int = 5)
    offsets = Int64[0; cumsum(map(length, layers))]
    flat = vcat(layers...)
    total = length(flat)
    centres = isempty(box_centres) ? C_NULL : hcat(box_centres...)      # 2 x total, column per entry
    iterations = zeros(Int32, total); f_calls = zeros(Int32, total); max_values = zeros(total); status = zeros(Int32, total)
    cfgc = Ref(COptimConfig(1e-4, 1.0, 50, include_kl, 1e-7, 1e-6, 1e-8, 1.0, 1e9, tr_secular_iters, 0))
    st = ccall((:celeste_joint_infer, libceleste), Cint,
               (Ptr{Void}, Ptr{Float64}, Int32, Ptr{Int64}, Ptr{Int32}, Ptr{Float64}, Ref{COptimConfig},
                Ptr{Int32}, Ptr{Int32}, Ptr{Float64}, Ptr{Int32}),
               ctx.handle, vp_all, length(layers), offsets, flat, centres, cfgc, iterations, f_calls, max_values, status)
    return st, iterations, f_calls, max_values, status
end

"""The optimiser's trust-region sub-problem on its own (what Optim.jl's NewtonTrustRegion solves per iteration):
H is 41 x 41 x n in the free parameters, g 41 x n, delta n.  For comparing single Newton steps of a Julia run
with the device (solver 0: as celeste_maximize_batch, 1: eigen-decomposition)."""
function tr_solve_batch(H::Array{Float64,3}, g::Matrix{Float64}, delta::Vector{Float64}; solver::Int = 0, device::Int = 0)
    n = length(delta)
    p = zeros(41, n); m = zeros(n); interior = zeros(Int32, n); fell_back = zeros(Int32, n)
    check(ccall((:celeste_tr_solve_batch, libceleste), Cint,
                (Cint, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int32, Int32, Ptr{Float64}, Ptr{Float64},
                 Ptr{Int32}, Ptr{Int32}),
                device, n, H, g, delta, solver, 0, p, m, interior, fell_back))
    p, m, interior, fell_back
end


# ---- one process, N devices: the N workers of one_node_single_infer / one_node_joint_infer (ParallelRun.jl:546-607, 302-397)
# behind ONE handle.  The library replicates the images on every member device, shards a call's targets by estimate_time,
# runs every member on its own worker thread and exchanges the per-source results with one ncclAllGather (RCCL over xGMI):
# a Julia caller reaches all the GPUs of the node through these three calls, no Distributed / MPI on the Julia side.

struct CGroupInfo                  # celeste_group_info_t
    n_members::Int32; n_devices::Int32; exchange::Int32; rccl_ranks::Int32
    devices::NTuple{16,Int32}
end

mutable struct MI355XGroup
    handle::Ptr{Void}
    n_members::Int
end

"""A device group over the whole box: `prob` is the celeste_problem_t of every catalogued source (the whole-box context of the
section above), `devices` the 0-based HIP ordinals (default: 0 .. n - 1)."""
function MI355XGroup(prob::CProblem, devices::Vector{Int32})
    check_abi()
    h = Ref{Ptr{Void}}(C_NULL)
    check(ccall((:celeste_group_create, libceleste), Cint, (Ref{CProblem}, Int32, Ptr{Int32}, Ref{Ptr{Void}}),
                prob, length(devices), devices, h))
    g = MI355XGroup(h[], length(devices))
    finalizer(g, x -> ccall((:celeste_group_destroy, libceleste), Void, (Ptr{Void},), x.handle))
    g
end

function group_info(g::MI355XGroup)
    gi = Ref(CGroupInfo(0, 0, 0, 0, ntuple(i -> Int32(0), 16)))
    check(ccall((:celeste_group_info, libceleste), Cint, (Ptr{Void}, Ref{CGroupInfo}), g.handle, gi))
    gi[]
end

"""
(collectives each member has enqueued, aborted?).  `aborted` = a member dropped out in front of a collective and the library
tore the communicators down: every call on the group returns status 7 (CELESTE_ERR_ABORTED) from then on -- finalize the
group and build a new one (the reference's try/catch around a source, ParallelRun.jl:389-396, is the per-source status; a
device dropping out has no counterpart there).
"""
function group_collectives(g::MI355XGroup)
    enq = zeros(Int64, g.n_members); ab = Ref{Int32}(0)
    check(ccall((:celeste_group_collectives, libceleste), Cint, (Ptr{Void}, Ptr{Int64}, Ref{Int32}), g.handle, enq, ab))
    enq, ab[] != 0
end

"""elbo_batch! over every device of the group: same arguments, same outputs (in the order of `targets0`)."""
function elbo_batch!(out::PinnedOutputs, g::MI355XGroup, vp_all::Matrix{Float64}, targets0::Vector{Int32};
                     flags::UInt32 = UInt32(7 | 32))
    ccall((:celeste_group_elbo_eval_batch, libceleste), Cint,
          (Ptr{Void}, Ptr{Float64}, Int32, Ptr{Int32}, UInt32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int64}, Ptr{Int32}),
          g.handle, vp_all, length(targets0), targets0, flags, out.v, out.d, out.h, out.counters, out.status)
end

"""one_node_single_infer's loop over every device of the group (ParallelRun.jl:546-607)."""
function maximize_batch!(g::MI355XGroup, vp_all::Matrix{Float64}, targets0::Vector{Int32};
                         vp_frozen_neighbors = C_NULL, box_centres = C_NULL, include_kl::Bool = true,
                         tr_secular_iters::Int = 5)
    n = length(targets0)
    iterations = zeros(Int32, n); f_calls = zeros(Int32, n); max_values = zeros(n); status = zeros(Int32, n)
    cfgc = Ref(COptimConfig(1e-4, 1.0, 50, include_kl, 1e-7, 1e-6, 1e-8, 1.0, 1e9, tr_secular_iters, 0))
    st = ccall((:celeste_group_maximize_batch, libceleste), Cint,
               (Ptr{Void}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int32, Ptr{Int32}, Ref{COptimConfig},
                Ptr{Int32}, Ptr{Int32}, Ptr{Float64}, Ptr{Int32}),
               g.handle, vp_all, vp_frozen_neighbors, box_centres, n, targets0, cfgc, iterations, f_calls, max_values, status)
    return st, iterations, f_calls, max_values, status
end

"""
one_node_joint_infer over every device of the group.  `batches` is partition_cyclades_dynamic's result as the reference has
it (partition.jl:173-236): batches[b][k] = the 0-based source ids of connected component k of batch b, in the order they are
optimised.  The components of a batch are sharded over the devices; rows are exchanged only in front of a batch in which a
device reads a row another device has written since the last exchange (at most once per batch; a group of one: once).
box_centres: 2 x (number of entries), a column per entry of vcat(vcat(batches...)...).  Per-entry outputs: entry fastest, then
sweep.
"""
function joint_infer!(g::MI355XGroup, vp_all::Matrix{Float64}, batches::Vector{Vector{Vector{Int32}}};
                      n_sweeps::Int = 3, box_centres = C_NULL, include_kl::Bool = true, tr_secular_iters::Int = 5)
    comps = vcat(batches...)
    batch_offsets = Int64[0; cumsum(map(length, batches))]
    comp_offsets = Int64[0; cumsum(map(length, comps))]
    flat = vcat(comps...)
    total = length(flat) * n_sweeps
    iterations = zeros(Int32, total); f_calls = zeros(Int32, total); max_values = zeros(total); status = zeros(Int32, total)
    exchanges = Ref{Int64}(0)
    cfgc = Ref(COptimConfig(1e-4, 1.0, 50, include_kl, 1e-7, 1e-6, 1e-8, 1.0, 1e9, tr_secular_iters, 0))
    st = ccall((:celeste_group_joint_infer, libceleste), Cint,
               (Ptr{Void}, Ptr{Float64}, Int32, Int32, Ptr{Int64}, Ptr{Int64}, Ptr{Int32}, Ptr{Float64}, Ref{COptimConfig},
                Ptr{Int32}, Ptr{Int32}, Ptr{Float64}, Ptr{Int32}, Ref{Int64}),
               g.handle, vp_all, n_sweeps, length(batches), batch_offsets, comp_offsets, flat, box_centres, cfgc,
               iterations, f_calls, max_values, status, exchanges)
    return st, iterations, f_calls, max_values, status, exchanges[]
end
