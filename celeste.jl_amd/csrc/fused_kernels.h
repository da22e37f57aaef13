// fused_kernels.h -- ElboMaximize.maximize! for a batch of targets in ONE persistent launch (optim_fused_kernel).
//
// The chained optimiser (celeste_abi.hip: work list -> pixel_kernel -> lift_kernel -> optim_step_kernel per Newton
// iteration, all targets in lock-step) is bound, for small batches, by the latencies of four dependent launches: a
// Cyclades layer of 80 targets (ParallelRun.jl:302-397) spent 182 us per iteration for ~60 us of work on its critical
// path, with 1000 of the chip's 1024 SIMDs idle during the 92 us of the trust-region step.  Here every target runs its
// own Newton iterations at its own pace inside one launch, as a dataflow over a device-side queue:
//
//   queue item r >= 0   "evaluate chunk record r": the 256 pixels of chunk ch of the j-th patch of target ti
//                       (chunk_desc[r]).  A workgroup (4 wavefronts) forms the target's per-image tables in LDS from its
//                       current parameters (prep_visit_values), each wavefront runs ONE 64-pixel iteration of the pixel
//                       loop (pixel_iter, the body pixel_kernel runs four times in a row), the wavefronts add their
//                       pixels' entries to the chunk's record in iteration order, and the record goes to HBM.
//   last record of a    The workgroup whose record completes a target's evaluation (arrival counter) runs the rest of the
//   target's evaluation iteration itself: lift_target (records -> value, 44-gradient, 44 x 44 Hessian, in LDS), optim_step_target
//                       (chain rule, accept / reject, trust-region sub-problem, next trial point), and queues the target's
//                       chunk records again -- or retires the target.
//   item EXIT           pushed once per workgroup when the last target retires.
//
// No workgroup ever waits for a particular other workgroup: a workgroup that holds nothing waits for its queue ticket, and
// tickets are filled by workgroups that are running.  The launch is therefore deadlock-free whatever part of the grid is
// resident.  Every wait is bounded (wall clock): a time-out sets an abort code that every waiting workgroup sees.
//
// Hand-offs between workgroups (guide section 6, Guideline 16, forms R1 / R2): everything one workgroup writes for another
// -- a target's row of vp, its OptState and saved Hessian, chunk records, queue items -- is stored write-through
// (agent-scope relaxed atomic stores = `global_store ... sc1`, stc<true>) and loaded past the L1 (ldc<true>); the storing
// wavefronts drain their stores (s_waitcnt vmcnt(0)) before the flag -- the queue item or the arrival counter -- is
// published.  No release / acquire fences (1.7-6.5 us each on this chip) are needed.
//
// Joint mode (optim_fused_kernel<true>, celeste_joint_infer): the whole schedule of one_node_joint_infer -- every
// (sweep, Cyclades batch, layer, source) entry, ParallelRun.jl:302-397 -- in ONE launch, as a dataflow over the entries:
// an entry may start when every earlier entry that wrote a row it reads (its own source's, its neighbours') or read the
// row it writes has finished.  That is exactly the order the layer-by-layer schedule guarantees, without its barriers: a
// layer no longer waits for its slowest source, a batch no longer waits for its longest chain.  Three more item kinds:
//   START e             the entry's dependencies are met: save the source's row, start its optimiser state
//                       (optim_init_values), queue the rendering of its neighbours
//   RENDER (e, g)       four of the value-kernel items of the entry's source, one per wavefront (value_pixels): the
//                       neighbours' light on the overlaps with the source's patches, from the neighbours' CURRENT rows;
//                       the last group queues the entry's first chunk records
//   (end of an entry)   the workgroup that ran the entry's last step restores the row if the entry failed, refreshes the
//                       source's per-image tables and shape derivatives (what prep_kernel / setup_kernel do between
//                       layers), then counts the entry off its successors and queues those that become ready.
// Every entry sees the parameter table the layered schedule shows it, so the results are bit-identical to it.
//
// Results are bit-identical to the chained path: same device functions (pixel_iter, prep_visit_values, source_geo_values,
// lift_target, optim_step_target, the non-inlined tri_tr_solve / eig_tr_solve), same 256-pixel chunk records summed in
// the same order (tests/test_gpu_fused.py).
#pragma once
#include "elbo_kernels.h"
#include "optim_kernels.h"

#define FQ_EMPTY (-1)      // queue slot not written yet (the array is memset to 0xFF before every launch)
#define FQ_EXIT (-2)
#define FQ_DIRECT0 (-3)    // item <= FQ_DIRECT0: target ti = FQ_DIRECT0 - item has no pixel to visit -- straight to its step
#define FUSED_NT 256
#ifndef FUSED_GATED
#define FUSED_GATED 1      // 1: a wavefront forms its pixels' record entries before it waits for its turn to add them (pixel_iter)
#endif
#define FUSED_WAVES (FUSED_NT / 64)

// q_ctl words
#define FQC_HEAD 0         // next ticket
#define FQC_TAIL 1         // next free queue position
#define FQC_LIVE 2         // targets still running
#define FQC_ABORT 3        // 0, or why the launch gave up: 1 wait timed out, 2 queue capacity exceeded
#define FQC_WORDS 8

struct FusedArgs {
    // the context's tables (immutable during the launch)
    const DevImage *images; const DevPatch *patches; const double *coefs; const uint8_t *bitmaps;
    const int64_t *nbr_off; const int32_t *nbr_idx; const int64_t *val_off; const double2 *val;
    const int64_t *nv_base; const int32_t *nbr_vis; const int2 *items; const SrcGeo *geo; const PriorDev *prior;
    const int32_t *vis_off; const int32_t *vis_img; const double *lg_sum; const int32_t *rec_off;
    int N, NC, K, M, CH, chunk_px;
    // the batch
    const int32_t *targets; int n_targets;
    double *vp;                    // n_sources x 44: the targets' rows move, every other row is frozen
    double *acc;                   // chunk records, 68 doubles each
    const int4 *chunk_desc;        // per record, two entries: {ti, j, chunk, target} {visit, image, 0, 0}
    const int2 *tgt_rec;           // per target: {first record, number of records}
    OptState *st; double *Hstate; double *Tstate; OptParams op; uint32_t flags;   // Tstate: TRI_STATE doubles per target (optim_step_target)
    double *Spec;                  // 2 x SPEC_STATE doubles per target: the quarter-radius step computed ahead (fused_speculate); nullptr: off
    // the queue
    int32_t *q_items; int32_t *q_ctl; int32_t *arrivals; int q_cap;
    long long timeout_ticks;       // wall_clock64 ticks (100 MHz) a workgroup waits for its ticket before it gives up
    // joint mode (optim_fused_kernel<true>): `targets` lists the entries of the schedule (sources repeat)
    int j_R;                       // number of chunk records = first START item; START e = j_R + e
    int j_gshift;                  // RENDER (e, g) = j_R + n_targets + (e << j_gshift | g)
    int32_t *j_dep;                // per entry: predecessors still running
    const int32_t *j_succ_off; const int32_t *j_succ;   // per entry: the entries that wait for it
    const int32_t *j_vitem_off; const int4 *j_vitems;   // per source: its value-kernel items (celeste_ctx_create)
    int32_t *j_render_arr;         // per entry: render groups done
    double *j_saved;               // per entry: the source's row before the entry
    const double *j_pos;           // per entry: centre of the position box (nullptr: the position at its start)
    SrcImg *j_srcimg; Comp *j_comps; SrcGeo *j_geo;     // the context's per-visit / per-source tables, refreshed in flight
};
// How the phases of the persistent kernel see the launch's arguments: in place, in the kernel-argument segment (constant
// address space: scalar loads through the scalar cache).  Passed as `const FusedArgs &` they were a copy on the kernel's
// stack -- 464 B of scratch per lane, every field a flat load from it in front of the access it serves.
typedef const __attribute__((address_space(4))) FusedArgs &FusedArgsK;
// (handed down as a plain pointer and made wave-uniform again inside every phase: a pointer that arrives as a function
// argument is a per-lane value to the compiler, and the loads through it would be vector loads)
__device__ __forceinline__ FusedArgsK fused_args(const FusedArgs *p) {
    const unsigned long long b = (unsigned long long)(size_t)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return *(const __attribute__((address_space(4))) FusedArgs *)(size_t)(((unsigned long long)hi << 32) | lo);
}


// per batch, once: the targets' record ranges, the description of every record, the first round of queue items
// (q_items == nullptr: no queue -- eval_fused_kernel)
__global__ void fused_setup_kernel(const int32_t *__restrict__ targets, int n_targets, const DevPatch *__restrict__ patches,
                                   const int32_t *__restrict__ vis_off, const int2 *__restrict__ items, int N, int M,
                                   int chunk_px, const int32_t *__restrict__ rec_off, int4 *__restrict__ chunk_desc,
                                   int2 *__restrict__ tgt_rec, int32_t *__restrict__ q_items, int32_t *__restrict__ q_ctl,
                                   const int32_t *__restrict__ j_dep = nullptr, int j_R = 0) {
    const int ti = blockIdx.x * blockDim.x + threadIdx.x;
    if (ti >= n_targets) return;
    const int t = targets[ti];
    int first;
    const int n_rec = describe_target_records(ti, t, patches, items, N, M, chunk_px, rec_off, chunk_desc, first);
    tgt_rec[ti] = make_int2(first < 0 ? 0 : first, n_rec);
    if (!q_items) return;
    if (j_dep) {     // joint mode: the entries without predecessors start
        if (j_dep[ti] == 0) q_items[atomicAdd(&q_ctl[FQC_TAIL], 1)] = j_R + ti;
        if (ti == 0) q_ctl[FQC_LIVE] = n_targets;
        return;
    }
    const int cnt = n_rec > 0 ? n_rec : 1;
    const int base = atomicAdd(&q_ctl[FQC_TAIL], cnt);
    if (n_rec > 0) for (int i = 0; i < n_rec; ++i) q_items[base + i] = first + i;
    else q_items[base] = FQ_DIRECT0 - ti;
    if (ti == 0) q_ctl[FQC_LIVE] = n_targets;
}

// LDS of a workgroup of the fused kernel
struct FusedShared {
    // evaluating a chunk record
    double theta[CEL_P];
    Comp tc[14 * CEL_MAXK];
    double tcx[COMPX * 14 * CEL_MAXK];
    SrcImg si;
    double etab[64];
    double sacc[ACC_N * ACC_SLOTS];
    int turn;                          // the wavefront whose pixels are added to the record next
    // hand-over between the phases of one workgroup
    int item, last, done;
    // the evaluation of a target, from its lift to its step
    double ev_v, ev_d[CEL_P], ev_h[CEL_P * CEL_P];
    int ev_status;
    LiftShared lift;
    StepShared step;
};

#ifdef FUSED_TIMING   // debug builds (tools/variants): shader clocks per phase of the fused kernel, thread 0 of every workgroup
__device__ unsigned long long g_fused_clk[16];
#define FT_DECL long long ft__ = clock64()
#define FT(k) do { const long long now__ = clock64(); if (threadIdx.x == 0) atomicAdd(&g_fused_clk[k], (unsigned long long)(now__ - ft__)); ft__ = clock64(); } while (0)
#define FT_COUNT(k) do { if (threadIdx.x == 0) atomicAdd(&g_fused_clk[k], 1ull); } while (0)
#else
#define FT_DECL do { } while (0)
#define FT(k) do { } while (0)
#define FT_COUNT(k) do { } while (0)
#endif

__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// thread 0: the next queue item of this workgroup (FQ_EXIT when the launch is over or has been aborted)
__device__ __forceinline__ int fused_pop(const FusedArgs *Ap) {
    FusedArgsK A = fused_args(Ap);
    // (the abort word is looked at while waiting, not here: one dependent round trip less on every pop)
    const int ticket = __hip_atomic_fetch_add(&A.q_ctl[FQC_HEAD], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (ticket >= A.q_cap) {
        __hip_atomic_store(&A.q_ctl[FQC_ABORT], 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return FQ_EXIT;
    }
    const long long t0 = wall_clock64();
    for (unsigned spins = 0;; ++spins) {
        const int v = ldc<true>(&A.q_items[ticket]);
        if (v != FQ_EMPTY) return v;
        __builtin_amdgcn_s_sleep(8);
        if ((spins & 63u) == 63u) {
            if (ldc<true>(&A.q_ctl[FQC_ABORT]) != 0) return FQ_EXIT;
            if (wall_clock64() - t0 > A.timeout_ticks) {
                __hip_atomic_store(&A.q_ctl[FQC_ABORT], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return FQ_EXIT;
            }
        }
    }
}

// the whole workgroup: append `count` items first, first + 1, ... (count > 0), or the one item `first` (count == 0), or
// `-count` copies of `first` (count < 0).  Every store of the calling workgroup that the items' consumers depend on must
// have been drained (drain_stores + __syncthreads) before.
__device__ __forceinline__ void fused_push(const FusedArgs *Ap, const int tid, int *s_base, int first, int count) {
    FusedArgsK A = fused_args(Ap);
    const int n = count > 0 ? count : (count == 0 ? 1 : -count);
    if (tid == 0) *s_base = __hip_atomic_fetch_add(&A.q_ctl[FQC_TAIL], n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int base = *s_base;
    if (base + n > A.q_cap) {
        if (tid == 0) __hip_atomic_store(&A.q_ctl[FQC_ABORT], 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else
        for (int i = tid; i < n; i += FUSED_NT) stc<true>(&A.q_items[base + i], count > 0 ? first + i : first);
    __syncthreads();
}

// the entry's evaluation items: its chunk records, or the direct item of a source that visits no pixel
__device__ __forceinline__ void fused_push_eval(const FusedArgs *Ap, const int tid, int *s_base, int ti) {
    FusedArgsK A = fused_args(Ap);
    const int2 tr = A.tgt_rec[ti];
    if (tr.y > 0) fused_push(Ap, tid, s_base, tr.x, tr.y);
    else fused_push(Ap, tid, s_base, FQ_DIRECT0 - ti, 0);
}

// ---- joint mode: the start of an entry ----
static __device__ __noinline__ void joint_start(const FusedArgs *Ap, FusedShared &F, const int tid, const int e) {
    FusedArgsK A = fused_args(Ap);
    const int t = A.targets[e];
    double *const row = A.vp + (size_t)t * CEL_P;
    OptState *const js = reinterpret_cast<OptState *>(F.ev_h);          // (LDS scratch: no evaluation is in flight here)
    if (tid < CEL_P) {
        const double v = ldc<true>(row + tid);
        F.theta[tid] = v;
        stc<true>(A.j_saved + (size_t)e * CEL_P + tid, v);
    }
    __syncthreads();
    if (tid == 0) optim_init_values(F.theta, A.op, A.j_pos ? A.j_pos + 2 * (size_t)e : nullptr, *js);
    __syncthreads();
    if (tid < CEL_P) stc<true>(row + tid, F.theta[tid]);
    {
        double *dst = reinterpret_cast<double *>(&A.st[e]);
        const double *src = reinterpret_cast<const double *>(js);
        static_assert(sizeof(OptState) % sizeof(double) == 0, "OptState is copied in 8-byte words");
        for (int i = tid; i < (int)(sizeof(OptState) / sizeof(double)); i += FUSED_NT) stc<true>(dst + i, src[i]);
    }
    drain_stores();
    __syncthreads();
    const int n_it = A.j_vitem_off[t + 1] - A.j_vitem_off[t];
    if (n_it > 0) fused_push(Ap, tid, &F.done, A.j_R + A.n_targets + (e << A.j_gshift), (n_it + FUSED_WAVES - 1) / FUSED_WAVES);
    else fused_push_eval(Ap, tid, &F.done, e);
}

// ---- joint mode: one group of value-kernel items of an entry's source, one item per wavefront ----
static __device__ __noinline__ void joint_render(const FusedArgs *Ap, FusedShared &F, const int tid, const int code) {
    FusedArgsK A = fused_args(Ap);
    const int lane = tid & 63, wave = tid >> 6;
    const int e = code >> A.j_gshift, g = code & ((1 << A.j_gshift) - 1);
    const int t = A.targets[e];
    const int i0 = A.j_vitem_off[t], i1 = A.j_vitem_off[t + 1];
    if (wave == 0) F.etab[lane] = g_exp2_table[lane];
    __syncthreads();
    const int idx = i0 + g * FUSED_WAVES + wave;
    if (idx < i1) {
        const int4 it = A.j_vitems[idx];
        const int sn = it.x, ch = it.z;
        const DevPatch &P = A.patches[sn];
        const DevPatch &T = A.patches[it.y];
        // (value_kernel's item: the overlap of the neighbour's patch, minus its last column, with the target's)
        const int h_lo = max(P.off_h, T.off_h), h_hi = min(P.off_h + P.H2, T.off_h + T.H2);
        const int w_lo = max(P.off_w, T.off_w), w_hi = min(P.off_w + P.W2 - 1, T.off_w + T.W2);
        const int RH = h_hi - h_lo, RW = w_hi - w_lo;
        const int npx = RH * RW, p0 = ch * A.chunk_px;
        if (RH > 0 && RW > 0 && p0 < npx) {
            // the neighbour's tables: another workgroup refreshed them when the neighbour's last entry ended
            Comp *const tc = reinterpret_cast<Comp *>(F.ev_h) + wave * (14 * CEL_MAXK);
            SrcImg *const siw = reinterpret_cast<SrcImg *>(F.sacc) + wave;
            {
                const double *src = reinterpret_cast<const double *>(A.j_comps + (size_t)sn * A.NC);
                double *dst = reinterpret_cast<double *>(tc);
                for (int i = lane; i < A.NC * 8; i += 64) dst[i] = ldc<true>(src + i);
                static_assert(sizeof(SrcImg) % sizeof(double) == 0 && sizeof(SrcImg) / sizeof(double) <= 64, "SrcImg is copied in 8-byte words");
                if (lane < (int)(sizeof(SrcImg) / sizeof(double)))
                    reinterpret_cast<double *>(siw)[lane] = ldc<true>(reinterpret_cast<const double *>(A.j_srcimg + sn) + lane);
            }
            wave_sync();
            const SrcImg si = *siw;
            value_pixels<true>(lane, P, si, tc, A.NC, A.coefs, F.etab, h_lo, w_lo, RH, p0, min(npx, p0 + A.chunk_px),
                               const_cast<double2 *>(A.val) + A.val_off[sn]);
        }
    }
    drain_stores();
    __syncthreads();
    if (tid == 0) {
        const int ng = (i1 - i0 + FUSED_WAVES - 1) / FUSED_WAVES;
        F.last = __hip_atomic_fetch_add(&A.j_render_arr[e], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ng - 1;
    }
    __syncthreads();
    const bool last = F.last != 0;
    __syncthreads();
    if (last) fused_push_eval(Ap, tid, &F.done, e);
}

// ---- joint mode: the end of an entry (its last step has been stored and drained) ----
static __device__ __noinline__ void joint_end(const FusedArgs *Ap, FusedShared &F, const int tid, const int e) {
    FusedArgsK A = fused_args(Ap);
    const int lane = tid & 63, wave = tid >> 6;
    const int t = A.targets[e];
    double *const row = A.vp + (size_t)t * CEL_P;
    // a source that failed keeps the row it had before the entry (ParallelRun.jl:582-597)
    const bool failed = ldc<true>(&A.st[e].status) != CELESTE_OK;
    if (tid < CEL_P) {
        const double v = failed ? ldc<true>(A.j_saved + (size_t)e * CEL_P + tid) : ldc<true>(row + tid);
        if (failed) stc<true>(row + tid, v);
        F.theta[tid] = v;
    }
    __syncthreads();
    // the source's per-image tables and shape derivatives from its final row: what the next layer's prep_kernel and
    // setup kernel would compute for it as somebody's neighbour
    for (int j = wave; j < A.M; j += FUSED_WAVES) {
        int v = t * A.N + j, n = j;
        if (A.items) { v = A.items[e * A.M + j].x; n = A.items[e * A.M + j].y; }
        if (v < 0) continue;
        const DevPatch &P = A.patches[v];
        if (P.H2 * P.W2 <= 0) continue;
        Comp *const tc = reinterpret_cast<Comp *>(F.ev_h) + wave * (14 * CEL_MAXK);
        SrcImg *const siw = reinterpret_cast<SrcImg *>(F.sacc) + wave;
        prep_visit_values<true>(lane, F.theta, P, A.images[n].band - 1, A.K, siw, tc);
        wave_sync();
        double *dst = reinterpret_cast<double *>(A.j_comps + (size_t)v * A.NC);
        const double *src = reinterpret_cast<const double *>(tc);
        for (int i = lane; i < A.NC * 8; i += 64) stc<true>(dst + i, src[i]);
        if (lane < (int)(sizeof(SrcImg) / sizeof(double)))
            stc<true>(reinterpret_cast<double *>(A.j_srcimg + v) + lane, reinterpret_cast<const double *>(siw)[lane]);
        wave_sync();
    }
    SrcGeo *const gl = reinterpret_cast<SrcGeo *>(F.etab);
    static_assert(sizeof(SrcGeo) <= 64 * sizeof(double) && sizeof(SrcGeo) % sizeof(double) == 0, "SrcGeo staged in the exp table's LDS");
    __syncthreads();
    if (tid == 0) { source_geo_values(F.theta, gl); gl->pad = 0; }
    __syncthreads();
    if (tid < (int)(sizeof(SrcGeo) / sizeof(double)))
        stc<true>(reinterpret_cast<double *>(A.j_geo + t) + tid, reinterpret_cast<const double *>(gl)[tid]);
    drain_stores();
    __syncthreads();
    // count the entry off its successors; those that become ready start
    int *const ready = reinterpret_cast<int *>(F.sacc);
    if (tid == 0) F.turn = 0;
    __syncthreads();
    const int s0 = A.j_succ_off[e], ns = A.j_succ_off[e + 1] - s0;
    for (int i = tid; i < ns; i += FUSED_NT) {
        const int s = A.j_succ[s0 + i];
        if (__hip_atomic_fetch_add(&A.j_dep[s], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1)
            ready[atomicAdd(&F.turn, 1)] = s;
    }
    __syncthreads();
    const int nready = F.turn;
    if (nready > 0) {
        if (tid == 0) F.done = __hip_atomic_fetch_add(&A.q_ctl[FQC_TAIL], nready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int base = F.done;
        if (base + nready > A.q_cap) {
            if (tid == 0) __hip_atomic_store(&A.q_ctl[FQC_ABORT], 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else
            for (int i = tid; i < nready; i += FUSED_NT) stc<true>(&A.q_items[base + i], A.j_R + ready[i]);
    }
    __syncthreads();
}

// one chunk record on the calling workgroup: the four 64-pixel iterations on its four wavefronts, the record stored, the
// target's arrival counted (F.last: this record completed the target's evaluation).  Returns the target's slot.
// NOT inlined: its register allocation is then the pixel loop's own (pixel_kernel's), whatever surrounds the call.
template <bool JOINT>
static __device__ __noinline__ int fused_chunk_record(const FusedArgs *Ap, FusedShared &F, const int tid, const int item) {
    FusedArgsK A = fused_args(Ap);
    const int lane = tid & 63, wave = tid >> 6;
    FT_DECL;
    const int4 d0 = A.chunk_desc[2 * item], d1 = A.chunk_desc[2 * item + 1];
    const int ti = d0.x;
    const int j = d0.y, ch = d0.z, t = d0.w, v = d1.x, n = d1.y;
    const DevPatch &P = A.patches[v];
    const int npx = P.H2 * P.W2;
    const int p0 = ch * A.chunk_px, p1 = min(npx, p0 + A.chunk_px);
    // the target's current parameters (another workgroup stepped it), then its tables for this image
    if (tid < CEL_P) F.theta[tid] = ldc<true>(A.vp + (size_t)t * CEL_P + tid);
    if (wave == 2) F.etab[lane] = g_exp2_table[lane];
    for (int i = tid; i < ACC_N * ACC_SLOTS; i += FUSED_NT) F.sacc[i] = 0.0;
    if (tid == 0) F.turn = 0;
    __syncthreads();
    if (wave == 0) {
        prep_visit_values<false>(lane, F.theta, P, A.images[n].band - 1, A.K, &F.si, F.tc);
        if (lane < A.NC) comp_extra(F.tc[lane], F.tcx + COMPX * lane);
    } else if (wave == 1) brightness_moments_wave(lane, F.theta, P, A.images[n].band - 1, &F.si);
    __syncthreads();
    FT(1);
    const int base = p0 + 64 * wave;
    if (base < p1) {
        PixWork<double> W;
        W.img = &A.images[n]; W.P = &P; W.patches = A.patches; W.bitmaps = A.bitmaps; W.nbr_idx = A.nbr_idx;
        W.nb0 = A.nbr_off[t]; W.nb1 = A.nbr_off[t + 1];
        W.nv = A.nbr_vis ? A.nbr_vis + A.nv_base[t] + (int64_t)j * (W.nb1 - W.nb0) : nullptr;
        W.val_off = A.val_off; W.val = A.val; W.active_rank = nullptr; W.my_rank = 0;
        W.N = A.N; W.n = n; W.NC = A.NC; W.v = v;
        W.si = F.si;
        W.tc = F.tc; W.tcx = F.tcx; W.tcr = reinterpret_cast<const CompR<double> *>(F.tc);
        W.etab = F.etab;
        W.tcoef = A.coefs + (size_t)(CELESTE_MUTANT == 3 ? 0 : P.stamp) * (CEL_COEF * CEL_COEF);
        W.tile_off = nullptr; W.rec = nullptr;
        double a[3] = {0.0, 0.0, 0.0};
        volatile int *turn = &F.turn;
        // pixel_kernel's wavefront adds iteration after iteration into the slots; here iteration w belongs to
        // wavefront w, and the wavefronts take turns in the same order
        pixel_iter<2, double, false, FUSED_GATED != 0, JOINT>(W, base, p1, lane, F.sacc + (lane & (ACC_SLOTS - 1)), a, [&]() {
            while (*turn != wave) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        });
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the adds have been performed ...
        if (lane == 0) *turn = wave + 1;                          // ... before the next wavefront starts its own
    }
    __syncthreads();
    FT(2);
    if (wave == 0) {
        double *out = A.acc + (size_t)item * ACC_N;
        fold_record_slots<2>(F.sacc, lane, [&](int e, double s) { stc<true>(out + e, s); });
        drain_stores();
        if (lane == 0) {
            const int before = __hip_atomic_fetch_add(&A.arrivals[ti], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            F.last = before == A.tgt_rec[ti].y - 1;
        }
    }
    __syncthreads();
    FT(3); FT_COUNT(14);
    return ti;
}

// the lift, then the Newton step (returns whether the target is done) of a target whose records are complete.
// NOT inlined: every phase of the persistent loop keeps its own register allocation.
static __device__ __noinline__ void fused_lift(const FusedArgs *Ap, FusedShared &F, const int tid, const int ti) {
    FusedArgsK A = fused_args(Ap);
    const int t = A.targets[ti];
    if (tid == 0) stc<true>(&A.arrivals[ti], 0);
    lift_target<true>(F.lift, tid, ti, t, A.vp, A.images, A.patches, A.geo, A.nbr_off, A.nbr_idx, A.acc, A.prior, A.vis_off,
                      A.vis_img, A.N, A.M, A.CH, A.chunk_px, A.flags, &F.ev_v, F.ev_d, F.ev_h, nullptr, &F.ev_status,
                      A.lg_sum, A.rec_off);
    __syncthreads();
}
static __device__ __noinline__ int fused_step(const FusedArgs *Ap, FusedShared &F, const int tid, const int ti) {
    FusedArgsK A = fused_args(Ap);
    const int t = A.targets[ti];
    return optim_step_target<true, FUSED_NT>(F.step, tid, A.st[ti], A.Hstate + (size_t)ti * NF * NF,
                                             A.vp + (size_t)t * CEL_P, F.ev_h, F.ev_d, -F.ev_v, F.ev_status, A.op,
                                             A.Tstate ? A.Tstate + (size_t)ti * TRI_STATE : nullptr,
                                             A.Spec ? A.Spec + (size_t)ti * 2 * SPEC_STATE : nullptr);
}

// While the trial point of target ti is being evaluated by other workgroups, the workgroup that queued it computes what
// a REJECTION of that point would ask for: the step from the same accepted point with a quarter of the radius (tri_step on
// the reduced form kept in Tstate).  If the point is indeed rejected (19 % of the iterations), the step is there
// (optim_step_target, `spec`): 66 -> 42 us for such an iteration.  Same code, same inputs as the step computed on
// demand, so the same bits; skipped while the queue is long (the workgroup has better things to do) and in the hard
// case.  The tag -- the iteration the step belongs to -- is written last.
static __device__ __noinline__ void fused_speculate(const FusedArgs *Ap, FusedShared &F, const int tid, const int ti) {
    FusedArgsK A = fused_args(Ap);
    if (tid == 0) {
        const int h = ldc<true>(&A.q_ctl[FQC_HEAD]), t = ldc<true>(&A.q_ctl[FQC_TAIL]);
        F.last = t - h <= (int)gridDim.x / 4;        // (HEAD runs ahead of TAIL by the idle workgroups' tickets when the queue is empty)
    }
    __syncthreads();
    const bool go = F.last != 0;
    __syncthreads();
    if (!go) return;
    StepShared &Z = F.step;
    OptState &S = A.st[ti];
    const double *const Ts = A.Tstate + (size_t)ti * TRI_STATE;
    for (int k = tid; k < NF * NF; k += FUSED_NT) { const int j = k / NF; Z.sA[(k - j * NF) + LDA * j] = ldc<true>(&Ts[k]); }
    if (tid < NF) Z.sx[tid] = ldc<true>(&S.x[tid]);
    __syncthreads();
    if (tid < 64) {
        const bool fr = tid < NF;
        const double *v = Ts + NF * NF;
        TriForm TF;
        TF.td = fr ? ldc<true>(v + tid) : 0.0; TF.te = fr ? ldc<true>(v + NF + tid) : 0.0;
        TF.hv = fr ? ldc<true>(v + 2 * NF + tid) : 0.0; TF.gt = fr ? ldc<true>(v + 3 * NF + tid) : 0.0;
        TF.wmin = ldc<true>(v + 4 * NF); TF.wmax = ldc<true>(v + 4 * NF + 1);
        TF.wmin_lower = ldc<true>(v + 4 * NF + 2); TF.norm_bound = ldc<true>(v + 4 * NF + 3);
        const double delta = 0.25 * Z.s_delta;                            // what a rejection makes of the radius (this workgroup's own step left it here)
        const TriLds L = {Z.sA, Z.sw, Z.sU, Z.se, Z.sU + NF, Z.sgt_q};  // (optim_step_target's layout)
        const TrResult R = tri_step(L, TF, delta, tid, A.op.secular_iters);
        double *const Sp = A.Spec + ((size_t)ti * 2 + (Z.s_iter & 1)) * SPEC_STATE;
        if (R.solved) {
            if (fr) stc<true>(Sp + tid, Z.sx[tid] + R.p);
            if (tid == 0) { stc<true>(Sp + NF, R.m); stc<true>(Sp + NF + 1, R.interior ? 1.0 : 0.0); }
            drain_stores();
            if (tid == 0) stc<true>(Sp + NF + 2, (double)Z.s_iter);       // (not read back from memory: the target may have moved on)
        }
    }
    __syncthreads();
}

// disable_tail_calls: nothing here is a tail call, but without the attribute the optimiser marks the calls of the phases
// `tail` (no stack object of this function escapes any more), and a function with such a caller is no longer eligible for
// the no-callee-saved-registers convention: every phase then saved and restored up to a hundred VGPRs through scratch.
template <bool JOINT>
__global__ void __launch_bounds__(FUSED_NT, 2) __attribute__((disable_tail_calls))
optim_fused_kernel(const FusedArgs A_) {
    (void)A_;   // (the launch's only argument: read where it lies, FusedArgsK)
    const FusedArgs *const Ap = (const FusedArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    FusedArgsK A = fused_args(Ap);
    __shared__ FusedShared F;
    for (;;) {
        // The thread index passes through an opaque asm in every trip: otherwise the compiler hoists everything that
        // depends only on it -- index arithmetic and per-thread tables of the lift and the step, hundreds of values -- out
        // of this loop and keeps it alive (in scratch) across the pixel code.
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wave = tid >> 6;
        FT_DECL;
        if (tid == 0) F.item = fused_pop(Ap);
        __syncthreads();
        const int item = F.item;
        if (item == FQ_EXIT) break;
        FT(0);
        int ti;
        bool last;
        if (JOINT && item >= A.j_R) {
            if (item < A.j_R + A.n_targets) joint_start(Ap, F, tid, item - A.j_R);
            else joint_render(Ap, F, tid, item - A.j_R - A.n_targets);
            continue;
        }
        if (item >= 0) {
            ti = fused_chunk_record<JOINT>(Ap, F, tid, item);
            last = F.last != 0;
        } else {
            ti = FQ_DIRECT0 - item;
            last = true;
        }
        if (!last) continue;

        // ---- the target's evaluation is complete: lift, Newton step ----
        fused_lift(Ap, F, tid, ti);
        FT(4);
        const int done = fused_step(Ap, F, tid, ti);
        FT(5);
        drain_stores();      // the target's row of vp, its state and saved Hessian are in memory ...
        __syncthreads();     // ... before its next items (or the end of the launch) become visible
        if (!done) {
            fused_push_eval(Ap, tid, &F.done, ti);
            if (A.Spec && A.op.solver != 1) fused_speculate(Ap, F, tid, ti);
        } else {
            if (JOINT) joint_end(Ap, F, tid, ti);
            if (tid == 0)
                F.done = __hip_atomic_fetch_add(&A.q_ctl[FQC_LIVE], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1;
            __syncthreads();
            const bool all_done = F.done != 0;
            __syncthreads();
            if (all_done) fused_push(Ap, tid, &F.done, FQ_EXIT, -(int)gridDim.x);
        }
        FT(6); FT_COUNT(15);
    }
}

// ---------------------------------------------------------------------------------------------------------
// eval_fused_kernel: one elbo() sweep of a SMALL batch (a rank's shard at N >= 4, a single call) with the fused kernel's
// chunk evaluation -- one workgroup per chunk record, its four 64-pixel iterations on four wavefronts adding in
// iteration order -- and the lift done by whichever workgroup completes a target (arrival counter), instead of
// pixel_kernel (one wavefront per record, four iterations in a row: the critical path of a batch that cannot fill the
// chip) followed by a lift launch.  Records and results are bit-identical to that pair's.  No queue and no waiting:
// every workgroup does its record, and at most one lift.  The per-image tables come from prep_kernel (HBM).
// ---------------------------------------------------------------------------------------------------------
struct EvalFusedShared {
    Comp tc[14 * CEL_MAXK];
    double tcx[COMPX * 14 * CEL_MAXK];
    double etab[64];
    double sacc[ACC_N * ACC_SLOTS];
    int turn, last;
    LiftShared lift;
};

__global__ void __launch_bounds__(FUSED_NT, 2)
eval_fused_kernel(const FusedArgs A, const SrcImg *__restrict__ srcimg, const Comp *__restrict__ comps,
                  const int32_t *__restrict__ n_records, int rec_bound, double *__restrict__ out_v, double *__restrict__ out_d,
                  double *__restrict__ out_h, int64_t *__restrict__ out_cnt, int32_t *__restrict__ out_status) {
    __shared__ EvalFusedShared F;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int item = blockIdx.x;
    if (item >= rec_bound) {
        // the last n_targets blocks: targets that visit no pixel at all (off every image) have no record to complete them
        const int te = item - rec_bound;
        if (A.tgt_rec[te].y > 0) return;
        const size_t HSe = (A.flags & CELESTE_FLAG_PACKED_HESS) ? CELESTE_HP : (size_t)CEL_P * CEL_P;
        lift_target<false, true>(F.lift, tid, te, A.targets[te], A.vp, A.images, A.patches, A.geo, A.nbr_off, A.nbr_idx, A.acc,
                                 A.prior, A.vis_off, A.vis_img, A.N, A.M, A.CH, A.chunk_px, A.flags, out_v + te,
                                 out_d + (size_t)te * CEL_P, out_h + (size_t)te * HSe, out_cnt ? out_cnt + 2 * (size_t)te : nullptr,
                                 out_status + te, A.lg_sum, A.rec_off);
        return;
    }
    if (item >= *n_records) return;              // rec_bound is the host's bound on the number of records
    const int4 d0 = A.chunk_desc[2 * item], d1 = A.chunk_desc[2 * item + 1];
    const int ti = d0.x, j = d0.y, ch = d0.z, t = d0.w, v = d1.x, n = d1.y;
    const DevPatch &P = A.patches[v];
    const int npx = P.H2 * P.W2;
    const int p0 = ch * A.chunk_px, p1 = min(npx, p0 + A.chunk_px);
    {
        const double *src = reinterpret_cast<const double *>(comps + (size_t)v * A.NC);
        double *dst = reinterpret_cast<double *>(F.tc);
        for (int i = tid; i < A.NC * 8; i += FUSED_NT) dst[i] = src[i];
        if (wave == 2) F.etab[lane] = g_exp2_table[lane];
        for (int i = tid; i < ACC_N * ACC_SLOTS; i += FUSED_NT) F.sacc[i] = 0.0;
        if (tid == 0) F.turn = 0;
        __syncthreads();
        if (tid < A.NC) comp_extra(F.tc[tid], F.tcx + COMPX * tid);
        __syncthreads();
    }
    const int base = p0 + 64 * wave;
    if (base < p1) {
        PixWork<double> W;
        W.img = &A.images[n]; W.P = &P; W.patches = A.patches; W.bitmaps = A.bitmaps; W.nbr_idx = A.nbr_idx;
        W.nb0 = A.nbr_off[t]; W.nb1 = A.nbr_off[t + 1];
        W.nv = A.nbr_vis ? A.nbr_vis + A.nv_base[t] + (int64_t)j * (W.nb1 - W.nb0) : nullptr;
        W.val_off = A.val_off; W.val = A.val; W.active_rank = nullptr; W.my_rank = 0;
        W.N = A.N; W.n = n; W.NC = A.NC; W.v = v;
        W.si = srcimg[v];
        W.tc = F.tc; W.tcx = F.tcx; W.tcr = reinterpret_cast<const CompR<double> *>(F.tc);
        W.etab = F.etab;
        W.tcoef = A.coefs + (size_t)(CELESTE_MUTANT == 3 ? 0 : P.stamp) * (CEL_COEF * CEL_COEF);
        W.tile_off = nullptr; W.rec = nullptr;
        double a[3] = {0.0, 0.0, 0.0};
        volatile int *turn = &F.turn;
        pixel_iter<2, double, false, FUSED_GATED != 0>(W, base, p1, lane, F.sacc + (lane & (ACC_SLOTS - 1)), a, [&]() {
            while (*turn != wave) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        });
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) *turn = wave + 1;
    }
    __syncthreads();
    if (wave == 0) {
        double *out = A.acc + (size_t)item * ACC_N;
        fold_record_slots<2>(F.sacc, lane, [&](int e, double s) { stc<true>(out + e, s); });
        drain_stores();
        if (lane == 0) {
            const int before = __hip_atomic_fetch_add(&A.arrivals[ti], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            F.last = before == A.tgt_rec[ti].y - 1;
            if (F.last) stc<true>(&A.arrivals[ti], 0);        // re-armed for the next launch
        }
    }
    __syncthreads();
    if (!F.last) return;
    const size_t HS = (A.flags & CELESTE_FLAG_PACKED_HESS) ? CELESTE_HP : (size_t)CEL_P * CEL_P;
    lift_target<false, true>(F.lift, tid, ti, t, A.vp, A.images, A.patches, A.geo, A.nbr_off, A.nbr_idx, A.acc, A.prior, A.vis_off,
                             A.vis_img, A.N, A.M, A.CH, A.chunk_px, A.flags, out_v + ti, out_d + (size_t)ti * CEL_P,
                             out_h + (size_t)ti * HS, out_cnt ? out_cnt + 2 * (size_t)ti : nullptr, out_status + ti, A.lg_sum,
                             A.rec_off);
}

// after the launch: per-target outputs from the optimiser states; a target that failed gets its input row back
// (ParallelRun.jl:582-597: the source is skipped, the others keep their results)
__global__ void optim_finalize_kernel(const OptState *__restrict__ st, const int32_t *__restrict__ targets, int n_targets,
                                      double *__restrict__ vp, const double *__restrict__ saved_rows,
                                      int32_t *__restrict__ iterations, int32_t *__restrict__ f_evals,
                                      double *__restrict__ elbo, int32_t *__restrict__ status,
                                      const int32_t *__restrict__ q_ctl) {
    const int ti = blockIdx.x * blockDim.x + threadIdx.x;
    if (ti >= n_targets) return;
    const OptState &S = st[ti];
    // q_ctl (fused launches): the launch gave up -- nothing it left behind is a result
    const int st1 = (q_ctl && q_ctl[FQC_ABORT] != 0) ? CELESTE_ERR_HIP : S.status;
    if (st1 != CELESTE_OK && saved_rows)
        for (int k = 0; k < CEL_P; ++k) vp[(size_t)targets[ti] * CEL_P + k] = saved_rows[(size_t)ti * CEL_P + k];
    if (iterations) iterations[ti] = S.iter;
    if (f_evals) f_evals[ti] = S.evals;
    if (elbo) elbo[ti] = -S.f;
    if (status) status[ti] = st1;
}

// rows of the targets before the optimisation (restored by optim_finalize_kernel for targets that fail)
__global__ void save_rows_kernel(const double *__restrict__ vp, const int32_t *__restrict__ targets, int n_targets,
                                 double *__restrict__ saved_rows) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_targets * CEL_P) return;
    const int ti = k / CEL_P, q = k - ti * CEL_P;
    saved_rows[k] = vp[(size_t)targets[ti] * CEL_P + q];
}
