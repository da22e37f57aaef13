// celeste_abi.hip -- host side of the C ABI declared in include/celeste_mi355x.h.
//
// Owns the device copies of the problem (images, patches, spline coefficients, neighbour
// graph) and launches prep_kernel -> pixel_kernel -> lift_kernel on one HIP stream.
// There is deliberately no CPU evaluation path: without a HIP device every entry point
// that computes returns CELESTE_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "elbo_kernels.h"
#include "optim_kernels.h"
#include "fused_kernels.h"

struct celeste_group;

// Evaluation batches of up to this many targets run eval_fused_kernel (launch_eval): a batch whose chunk records fit the
// chip in one round of 256-thread workgroups is latency-bound, and four wavefronts per record + the lift in the same
// launch shorten its critical path.  Beyond that the chip is full and pixel_kernel's one wavefront per record (its
// prologue paid once per four iterations, no turn-taking) has 2-3 x the throughput: a rank's N = 8 shard of the bench
// field (250 targets) takes 0.168 ms with pixel_kernel + lift_kernel, 0.287 ms with eval_fused_kernel (measured).
#define EVAL_FUSED_MAX 32

#define HIP_TRY(expr)                                                                    \
    do {                                                                                 \
        hipError_t e__ = (expr);                                                         \
        if (e__ != hipSuccess) {                                                         \
            fprintf(stderr, "celeste_mi355x: %s failed: %s (%s:%d)\n", #expr,            \
                    hipGetErrorString(e__), __FILE__, __LINE__);                         \
            return CELESTE_ERR_HIP;                                                      \
        }                                                                                \
    } while (0)

// Every device allocation of the library goes through here.  CELESTE_POISON=1 (testing) fills fresh allocations with 0xFF
// bytes -- NaN as a double or float, -1 as an integer -- so that a kernel that reads what no kernel or copy has written
// shows it in every run instead of in the rare one where recycled memory happens to hold something harmful (the tables of a
// batch are marked with batch stamps, not cleared: DESIGN.md section 3).
static hipError_t celeste_device_malloc(void **p, size_t bytes) {
    hipError_t e = (hipMalloc)(p, bytes);
    static const bool poison = [] { const char *v = getenv("CELESTE_POISON"); return v && atoi(v) != 0; }();
    if (e == hipSuccess && poison) {
        e = hipMemset(*p, 0xFF, bytes);
        if (e == hipSuccess) e = hipDeviceSynchronize();
    }
    return e;
}
#define hipMalloc(p, bytes) celeste_device_malloc((void **)(p), (bytes))

// ---- streams are recycled, never destroyed ------------------------------------------------------------------------------
// The HIP runtime (ROCm 7.0.2's libamdhip64, the one PyTorch bundles) was caught writing into a stream object AFTER
// hipStreamDestroy had freed it: with the host's cores oversubscribed (eight test processes, each with a 256-thread OpenMP
// checker) one context in ~800 left a stale decrement and a stale 32-bit zero in whatever heap block took the 920 bytes of its
// copy stream next -- a numpy array of the following test (found with tools/heapwho: the block's last owner allocated in
// hipStreamCreateWithFlags under celeste_ctx_create_on and freed in hipStreamDestroy under celeste_ctx_destroy; the streams
// were idle and synchronised, their events destroyed).  A library whose callers create a context per source
// (ParallelRun.jl:468-488) cannot afford a destroy that corrupts the caller's heap, so a context's streams go back to a
// per-device pool (idle, synchronised) and the next context takes them from there; the pool is never torn down.  It also
// keeps the stream -> hardware-queue assignment of a process stable (pick_copy_stream).  An idle stream holds about 1 MB of
// device memory (measured: 200 streams, 216 MB), so the pool keeps at most CELESTE_STREAM_POOL_MAX streams per device (default
// 32: the streams of 16 contexts); beyond that a stream is destroyed as before -- only a process that closes more than
// 16 contexts without opening one in between gets there.
struct StreamPool {
    std::mutex mu;
    std::vector<hipStream_t> idle[16];
};
static StreamPool &stream_pool() { static StreamPool *p = new StreamPool(); return *p; }   // (leaked on purpose: no HIP call at exit)
static hipError_t stream_acquire(int device, hipStream_t *out) {
    *out = nullptr;
    if (device >= 0 && device < 16) {
        StreamPool &sp = stream_pool();
        std::lock_guard<std::mutex> lk(sp.mu);
        if (!sp.idle[device].empty()) { *out = sp.idle[device].back(); sp.idle[device].pop_back(); return hipSuccess; }
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}
static void stream_retire(int device, hipStream_t s) {
    if (!s) return;
    if (hipStreamSynchronize(s) == hipSuccess && device >= 0 && device < 16) {
        static const size_t cap = [] { const char *e = getenv("CELESTE_STREAM_POOL_MAX"); return e ? (size_t)std::max(0, atoi(e)) : (size_t)32; }();
        StreamPool &sp = stream_pool();
        std::lock_guard<std::mutex> lk(sp.mu);
        if (sp.idle[device].size() < cap) { sp.idle[device].push_back(s); return; }
    } else (void)hipGetLastError();
    (void)hipStreamDestroy(s);
}

// ---- page-locked staging blocks are recycled too -------------------------------------------------------------------------
// A context's staging (the block pair of the small host-pointer calls, the parts' blocks of a host-pointer sweep, the
// optimiser's table and state copies) is page-locked memory, and the runtime takes 30 .. 115 us to lock a block and 210 .. 330 us
// to release one: of the 1.1 ms that creating a per-source context, calling it once and destroying it took (process_source's
// pattern, ParallelRun.jl:468-488; tools/gpu_per_source_ctx_time.py under rocprofv3 --hip-trace), two hipHostFree were 483 us and
// two hipHostMalloc 102 -- against 76 us for an evaluation.  Blocks of up to 1 MB go back to a pool of power-of-two size classes
// (CELESTE_PINNED_POOL_KB, default 16384 per process) instead; larger ones and everything beyond the cap are released as before.
struct PinnedPool {
    std::mutex mu;
    std::unordered_map<void *, int> live;       // blocks handed out -> size class
    std::vector<void *> idle[9];                // 4 KB << class, class 0 .. 8 (1 MB)
    size_t idle_bytes = 0;
};
static PinnedPool &pinned_pool() { static PinnedPool *p = new PinnedPool(); return *p; }   // (leaked on purpose)
static size_t pinned_pool_cap() {
    static const size_t cap = [] { const char *v = getenv("CELESTE_PINNED_POOL_KB"); return (size_t)(v ? std::max(0, atoi(v)) : 16384) << 10; }();
    return cap;
}
static hipError_t staging_alloc(void **p, size_t bytes) {
    *p = nullptr;
    int k = -1;
    if (pinned_pool_cap() && bytes <= (1u << 20)) { k = 0; while (((size_t)4096 << k) < bytes) ++k; }
    // (hipHostMallocPortable: a recycled block may serve a context on another device of the process)
    if (k < 0) return hipHostMalloc(p, std::max<size_t>(bytes, 1), hipHostMallocPortable);
    PinnedPool &pp = pinned_pool();
    {
        std::lock_guard<std::mutex> lk(pp.mu);
        if (!pp.idle[k].empty()) { *p = pp.idle[k].back(); pp.idle[k].pop_back(); pp.idle_bytes -= (size_t)4096 << k; pp.live[*p] = k; return hipSuccess; }
    }
    hipError_t e = hipHostMalloc(p, (size_t)4096 << k, hipHostMallocPortable);
    if (e == hipSuccess) { std::lock_guard<std::mutex> lk(pp.mu); pp.live[*p] = k; }
    return e;
}
static void staging_free(void *p) {
    if (!p) return;
    {
        PinnedPool &pp = pinned_pool();
        std::lock_guard<std::mutex> lk(pp.mu);
        auto it = pp.live.find(p);
        if (it != pp.live.end()) {
            const int k = it->second;
            pp.live.erase(it);
            if (pp.idle_bytes + ((size_t)4096 << k) <= pinned_pool_cap()) { pp.idle[k].push_back(p); pp.idle_bytes += (size_t)4096 << k; return; }
        }
    }
    (void)hipHostFree(p);
}

static const celeste_prior_t DEFAULT_PRIOR =
#include "prior_tables.inc"
    ;

// image planes in HBM, shared by every context created on the handle (reference counted)
struct celeste_images {
    int device = 0;
    int N = 0;
    std::vector<DevImage> h_images;
    std::vector<void *> plane_allocs;
    DevImage *d_images = nullptr;
    std::atomic<int> refs{1};
};

static void images_release(celeste_images *im) {
    if (!im || im->refs.fetch_sub(1) != 1) return;
    (void)hipSetDevice(im->device);
    for (void *p : im->plane_allocs) (void)hipFree(p);
    if (im->d_images) (void)hipFree(im->d_images);
    delete im;
}

struct celeste_ctx {
    int device = 0;
    celeste_images *imgs = nullptr;
    hipStream_t stream = nullptr;        // private non-blocking stream of the host-pointer entry points
    hipStream_t copy_stream = nullptr;   // device-to-host copies of finished parts of a batch
    bool copy_stream_checked = false;    // pick_copy_stream has run (first batch that overlaps copies with kernels)
    int N = 0, S = 0, K = 0, NC = 0, n_stamps = 0;
    int chunk_px = 256, CH = 1;
    int max_npx = 0;
    // host mirrors (for work stats / validation)
    std::vector<DevPatch> h_patches;
    std::vector<int64_t> h_nbr_off;
    std::vector<int32_t> h_nbr_idx;
    // device
    DevImage *d_images = nullptr;        // = imgs->d_images
    DevPatch *d_patches = nullptr;
    double *d_coefs = nullptr;
    float *d_coefs_f = nullptr;          // the spline coefficients rounded to float (single-precision mode)
    uint8_t *d_bitmaps = nullptr;
    int64_t *d_nbr_off = nullptr;
    int32_t *d_nbr_idx = nullptr;
    PriorDev *d_prior = nullptr;
    SrcImg *d_srcimg = nullptr;
    Comp *d_comps = nullptr;
    SrcGeo *d_geo = nullptr;
    int64_t *d_val_off = nullptr;   // per (source, image): offset of the patch in d_val
    double2 *d_val = nullptr;       // pre-rendered (E_G_s.v, var_G_s.v) of neighbour sources, per patch pixel
    int32_t *d_needed = nullptr;    // per source: stamp of the last batch it was a target of
    int32_t stamp = 0;
    // visit lists: the images each source has a non-empty patch in (grids run over these, tables stay S x N)
    std::vector<int32_t> h_vis_off, h_vis_img, h_vis_src;
    int32_t *d_vis_off = nullptr, *d_vis_img = nullptr, *d_vis_src = nullptr;
    int4 *d_value_items = nullptr;  // value_kernel work items: {neighbour's table entry, target's, chunk, target}
    int4 *d_vitems_src = nullptr;   // the same, grouped by target
    int32_t *d_vitem_off = nullptr; // S + 1 offsets into d_vitems_src
    std::vector<int32_t> h_vitem_off;
    int64_t n_value_items = 0;
    int32_t *d_prep_mark = nullptr; // per source: stamp of the last batch that read its per-image tables
    double *d_lg_sum = nullptr;     // per visit: sum of lgamma(pixel + 1) over the patch's visited pixels
    int64_t *d_nv_base = nullptr;   // visit-list mode: per source, start of its rows in d_nbr_vis
    int32_t *d_nbr_vis = nullptr;   // visit-list mode: [visit of s][neighbour of s] -> the neighbour's visit in that image
    int2 *d_items = nullptr;         // [ti * M + j] of the current batch: {visit, image}
    size_t items_cap = 0;
    // work list of pixel_kernel (work_count / work_scan / work_fill kernels), per batch
    std::vector<int32_t> h_src_chunks;   // per source: chunks of all its patches
    std::vector<int64_t> h_top_chunks;   // [k]: chunks of the k sources that have most (bound for k unknown targets)
    int max_src_chunks = 0;
    int32_t *d_work = nullptr, *d_work_blk = nullptr, *d_work_total = nullptr;
    size_t work_cap = 0, work_blk_cap = 0;
    int M = 1;        // largest number of images one source appears in
    bool dense = false;
    int64_t V = 0;    // visits in total
    // split variant: per (source, image) first 64-pixel tile in d_rec, allocated on first use
    std::vector<int64_t> h_tile_off;
    int64_t *d_tile_off = nullptr;
    int64_t n_tiles = 0;
    int sum_tiles = 4, RCH = 1;   // record_sum_kernel: tiles per workgroup, parts per patch
    double *d_rec = nullptr;
    double *d_acc_split = nullptr;
    size_t acc_split_cap = 0;
    // per-batch scratch (grown on demand)
    double *d_acc = nullptr;
    size_t acc_cap = 0;
    int32_t *d_rec_off = nullptr;    // [ti * M + j]: first chunk record of the j-th visit of target ti in d_acc
    size_t rec_off_cap = 0;
    // staging for the host-pointer API: device side, and page-locked host side
    double *d_vp = nullptr;
    int32_t *d_targets = nullptr;
    double *d_v = nullptr, *d_d = nullptr, *d_h = nullptr;
    int64_t *d_cnt = nullptr;
    int32_t *d_status = nullptr;
    size_t stage_cap = 0;
    double *p_vp = nullptr;              // pinned: S x 44
    int32_t *p_targets = nullptr, *p_status = nullptr;
    double *p_v = nullptr, *p_d = nullptr, *p_h = nullptr;
    int64_t *p_cnt = nullptr;
    size_t pin_cap = 0;
    // small host-pointer calls (eval_small): one block up (table + targets), one block down (all outputs)
    double *d_small_in = nullptr, *p_small_in = nullptr, *d_small_out = nullptr, *p_small_out = nullptr;
    static const int MAX_PARTS = 8;
    hipEvent_t part_done[MAX_PARTS] = {}, part_copied[MAX_PARTS] = {};
    // buffers of celeste_maximize_batch, kept between calls (grown on demand)
    struct OptBuffers {
        size_t cap = 0;
        double *d_vp = nullptr, *d_v = nullptr, *d_d = nullptr, *d_h = nullptr, *d_H = nullptr, *d_T = nullptr, *d_S = nullptr, *d_pos = nullptr;
        int32_t *d_targets = nullptr, *d_act[2] = {nullptr, nullptr}, *d_evt[2] = {nullptr, nullptr}, *d_count = nullptr,
                *d_st = nullptr;
        void *d_state = nullptr;
        // page-locked: the parameter table (in / out), the final optimiser states, a ring of live-target counts
        static const int RING = 4;
        double *h_vp = nullptr;
        void *h_state = nullptr;
        int32_t *h_count = nullptr;
        hipEvent_t ev[RING] = {};
    } opt;
    // buffers of the fused optimiser launch (optim_fused_kernel), kept between calls (grown on demand)
    struct FusedBuffers {
        size_t cap_t = 0, cap_rec = 0, cap_q = 0, cap_saved = 0;
        bool arrivals_dirty = true;        // the arrival counters may hold anything (fresh allocation, an aborted launch)
        int4 *d_chunk_desc = nullptr;
        int2 *d_tgt_rec = nullptr;
        int32_t *d_q_items = nullptr, *d_q_ctl = nullptr, *d_arrivals = nullptr;
        double *d_saved = nullptr;          // the targets' rows before the optimisation
        int32_t *h_ctl = nullptr;           // page-locked copy of the queue control words of the last launch
        int max_resident = 0;               // workgroups of optim_fused_kernel the device holds at once
        int cus = 0;                        // compute units of the device
    } fused;
    // device scratch of the less travelled entry points (eval_multi, render_expected): grown on demand, kept
    struct Scratch { void *p = nullptr; size_t cap = 0; } scratch[24];   // 13..22: celeste_joint_infer
    // timing
    int timing = 0;
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    int ev_split = 0;
    int ev_valid = 0;
};

// No C++ exception crosses the C ABI (the callers are C, Julia's ccall, ctypes): every entry point that can allocate through
// the standard library is a function-try-block ending in this handler.
#define ABI_CATCH catch (const std::bad_alloc &) { return CELESTE_ERR_ALLOC; } catch (...) { return CELESTE_ERR_HIP; }

extern "C" int celeste_version(void) { return CELESTE_ABI_VERSION; }

extern "C" const char *celeste_strerror(int status) {
    switch (status) {
        case CELESTE_OK: return "ok";
        case CELESTE_ERR_INVALID_ARG: return "invalid argument";
        case CELESTE_ERR_NONFINITE_INPUT: return "vp contains NaNs or Infs";
        case CELESTE_ERR_NONFINITE_RESULT: return "ELBO value, gradient or Hessian contains Inf/NaNs";
        case CELESTE_ERR_HIP: return "HIP runtime error";
        case CELESTE_ERR_NO_DEVICE: return "no HIP device (the engine has no CPU fallback)";
        case CELESTE_ERR_ALLOC: return "allocation failed";
        case CELESTE_ERR_ABORTED: return "device group aborted: a member failed in front of a collective (destroy the group and create it again)";
        default: return "unknown status";
    }
}

// ---- galaxy prototypes (light_source_model.jl:45-75) ----------------------------------------
static void galaxy_prototypes(double eta[16], double nu[16]) {
    const double dev_amp[8] = {4.26347652e-2, 2.40127183e-1, 6.85907632e-1, 1.51937350,
                               2.83627243, 4.46467501, 5.72440830, 5.60989349};
    const double dev_var[8] = {2.23759216e-4, 1.00220099e-3, 4.18731126e-3, 1.69432589e-2,
                               6.84850479e-2, 2.87207080e-1, 1.33320254, 8.40215071};
    const double exp_amp[6] = {2.34853813e-3, 3.07995260e-2, 2.23364214e-1, 1.17949102, 4.33873750, 5.99820770};
    const double exp_var[6] = {1.20078965e-3, 8.84526493e-3, 3.91463084e-2, 1.39976817e-1, 4.60962500e-1, 1.50159566};
    const double er0 = 1.078031, er1 = 0.928896;
    double sd = 0, se = 0;
    for (double a : dev_amp) sd += a;
    for (double a : exp_amp) se += a;
    for (int j = 0; j < 16; ++j) { eta[j] = 0; nu[j] = 0; }
    for (int j = 0; j < 8; ++j) { eta[j] = dev_amp[j] / sd; nu[j] = dev_var[j] / (er0 * er0); }
    for (int j = 0; j < 6; ++j) { eta[8 + j] = exp_amp[j] / se; nu[8 + j] = exp_var[j] / (er1 * er1); }
}

// ---- spline prefilter (imaged_sources.jl:97-107; Interpolations BSpline(Cubic(Line())), OnGrid) ----
// 1-D: n samples -> n + 2 coefficients.  The two boundary rows c[0] - 2 c[1] + c[2] = 0 reduce the
// first/last interior equations to c[1] = d[0], c[n] = d[n-1]; the rest is a tridiagonal
// (1, 4, 1) / 6 system solved with the Thomas algorithm.
static void prefilter_line(int n, const double *d, int dstride, double *c, int cstride) {
    std::vector<double> cp(n), dp(n), x(n + 2);
    x[1] = d[0];
    x[n] = d[(size_t)(n - 1) * dstride];
    const int m = n - 2;  // unknowns x[2..n-1]
    if (m > 0) {
        // (x[q-1] + 4 x[q] + x[q+1]) = 6 d[q-1], q = 2..n-1
        for (int q = 0; q < m; ++q) {
            double rhs = 6.0 * d[(size_t)(q + 1) * dstride];
            if (q == 0) rhs -= x[1];
            if (q == m - 1) rhs -= x[n];
            const double lower = (q == 0) ? 0.0 : 1.0;
            const double denom = 4.0 - lower * (q ? cp[q - 1] : 0.0);
            cp[q] = 1.0 / denom;
            dp[q] = (rhs - lower * (q ? dp[q - 1] : 0.0)) / denom;
        }
        x[2 + m - 1] = dp[m - 1];
        for (int q = m - 2; q >= 0; --q) x[2 + q] = dp[q] - cp[q] * x[2 + q + 1];
    }
    x[0] = 2.0 * x[1] - x[2];
    x[n + 1] = 2.0 * x[n] - x[n - 1];
    for (int q = 0; q < n + 2; ++q) c[(size_t)q * cstride] = x[q];
}

extern "C" int celeste_spline_prefilter(const double *stamp51, double *coef53) try {
    if (!stamp51 || !coef53) return CELESTE_ERR_INVALID_ARG;
    const int n = CEL_STAMP, m = CEL_COEF;
    std::vector<double> g((size_t)n * n), tmp((size_t)m * n);
    double sum = 0;
    for (int k = 0; k < n * n; ++k) { g[k] = std::fmax(stamp51[k], 0.0) + 1e-6; sum += g[k]; }
    for (int k = 0; k < n * n; ++k) {
        const double x = g[k] / sum;
        g[k] = (1000 * x > 1) ? 1000 * x - 1 : std::log(1000 * x);  // softpluslike (fsm_util.jl:221)
    }
    for (int w = 0; w < n; ++w) prefilter_line(n, g.data() + (size_t)n * w, 1, tmp.data() + (size_t)m * w, 1);
    for (int h = 0; h < m; ++h) prefilter_line(n, tmp.data() + h, m, coef53 + h, m);
    return CELESTE_OK;
} ABI_CATCH

static void inv4_logdet(const double *S, double *Inv, double *logdet) {
    double a[4][8];
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) { a[r][c] = S[r + 4 * c]; a[r][4 + c] = (r == c); }
    double det = 1;
    for (int c = 0; c < 4; ++c) {
        int piv = c;
        for (int r = c + 1; r < 4; ++r) if (std::fabs(a[r][c]) > std::fabs(a[piv][c])) piv = r;
        if (piv != c) { for (int k = 0; k < 8; ++k) std::swap(a[c][k], a[piv][k]); det = -det; }
        det *= a[c][c];
        const double inv = 1 / a[c][c];
        for (int k = 0; k < 8; ++k) a[c][k] *= inv;
        for (int r = 0; r < 4; ++r) if (r != c) { const double f = a[r][c]; for (int k = 0; k < 8; ++k) a[r][k] -= f * a[c][k]; }
    }
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) Inv[r + 4 * c] = a[r][4 + c];
    *logdet = std::log(det);
}

// The small tables of a context go up through ONE page-locked arena with asynchronous copies on the NULL stream (a synchronous
// hipMemcpy from pageable memory is 14 us, and a per-source context makes thirteen of them: 180 of its 300 us); celeste_ctx_create_on
// opens the arena (ArenaScope), the copies complete at its final wait for the NULL stream.  A table that does not fit (a whole
// field's patch list, 8765 PSF stamps) is copied synchronously as before.
struct UploadArena { char *base = nullptr; size_t cap = 0, used = 0; };
static thread_local UploadArena *tl_arena = nullptr;
struct ArenaScope {
    UploadArena a;
    explicit ArenaScope(size_t cap) {
        if (staging_alloc((void **)&a.base, cap) == hipSuccess) a.cap = cap;
        else { (void)hipGetLastError(); a.base = nullptr; }
        tl_arena = &a;
    }
    ~ArenaScope() {
        tl_arena = nullptr;
        if (a.base) { (void)hipStreamSynchronize(nullptr); staging_free(a.base); }   // (no copy out of the block is in flight when it goes back)
    }
    ArenaScope(const ArenaScope &) = delete;
    ArenaScope &operator=(const ArenaScope &) = delete;
};

template <class T>
static int dev_upload(T **dst, const T *src, size_t n) {
    *dst = nullptr;
    HIP_TRY(hipMalloc((void **)dst, std::max<size_t>(n, 1) * sizeof(T)));   // never a null table, even when empty
    if (src && n > 0) {
        const size_t bytes = n * sizeof(T);
        UploadArena *a = tl_arena;
        if (a && a->base && a->used + bytes <= a->cap) {
            memcpy(a->base + a->used, src, bytes);
            HIP_TRY(hipMemcpyAsync(*dst, a->base + a->used, bytes, hipMemcpyHostToDevice, nullptr));
            a->used += (bytes + 63) & ~(size_t)63;
        } else HIP_TRY(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
    }
    return CELESTE_OK;
}

static int select_device(int device) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) return CELESTE_ERR_NO_DEVICE;
    if (device < 0 || device >= count) return CELESTE_ERR_INVALID_ARG;
    HIP_TRY(hipSetDevice(device));
    return CELESTE_OK;
}

extern "C" int celeste_images_create(int32_t n_images, const celeste_image_t *images, int device,
                                     celeste_images_t **out) try {
    if (!images || !out || n_images <= 0) return CELESTE_ERR_INVALID_ARG;
    *out = nullptr;
    int st = select_device(device);
    if (st != CELESTE_OK) return st;
    celeste_images *c = new (std::nothrow) celeste_images();
    if (!c) return CELESTE_ERR_ALLOC;
    c->device = device; c->N = n_images;
#define IMG_TRY(expr) do { int s__ = (expr); if (s__ != CELESTE_OK) { images_release(c); return s__; } } while (0)
    c->h_images.resize(c->N);
    for (int n = 0; n < c->N; ++n) {
        const celeste_image_t &im = images[n];
        if (im.H <= 0 || im.W <= 0 || im.band < 1 || im.band > 5 || !im.pixels || !im.sky || !im.nelec_per_nmgy) {
            images_release(c); return CELESTE_ERR_INVALID_ARG;
        }
        DevImage d; d.H = im.H; d.W = im.W; d.band = im.band; d.pad = 0;
        float *dp = nullptr, *ds = nullptr, *di = nullptr;
        IMG_TRY(dev_upload(&dp, im.pixels, (size_t)im.H * im.W)); c->plane_allocs.push_back(dp);
        IMG_TRY(dev_upload(&ds, im.sky, (size_t)im.H * im.W)); c->plane_allocs.push_back(ds);
        IMG_TRY(dev_upload(&di, im.nelec_per_nmgy, (size_t)im.H)); c->plane_allocs.push_back(di);
        d.pixels = dp; d.sky = ds; d.iota = di;
        double *dli = nullptr;
        IMG_TRY(dev_upload<double>(&dli, nullptr, (size_t)im.H)); c->plane_allocs.push_back(dli);
        hipLaunchKernelGGL(iota_kernel, dim3((unsigned)((im.H + 255) / 256)), dim3(256), 0, nullptr, di, im.H, dli);
        d.log_iota = dli;
        c->h_images[n] = d;
    }
    IMG_TRY(dev_upload(&c->d_images, c->h_images.data(), c->h_images.size()));
#undef IMG_TRY
    if (hipDeviceSynchronize() != hipSuccess) { images_release(c); return CELESTE_ERR_HIP; }
    *out = c;
    return CELESTE_OK;
} ABI_CATCH

extern "C" void celeste_images_destroy(celeste_images_t *images) { images_release(images); }

extern "C" int celeste_ctx_create(const celeste_problem_t *pr, int device, celeste_ctx_t **out) try {
    if (!pr || !out) return CELESTE_ERR_INVALID_ARG;
    *out = nullptr;
    if (pr->n_images <= 0 || !pr->images) return CELESTE_ERR_INVALID_ARG;
    celeste_images_t *im = nullptr;
    int st = celeste_images_create(pr->n_images, pr->images, device, &im);
    if (st != CELESTE_OK) return st;
    st = celeste_ctx_create_on(im, pr, out);
    images_release(im);   // the context holds its own reference
    return st;
} ABI_CATCH

extern "C" int celeste_ctx_create_on(celeste_images_t *imgs, const celeste_problem_t *pr, celeste_ctx_t **out) try {
    if (!imgs || !pr || !out) return CELESTE_ERR_INVALID_ARG;
    *out = nullptr;
    if (pr->n_images != imgs->N || pr->n_sources <= 0 || pr->psf_K <= 0 || pr->psf_K > CEL_MAXK ||
        !pr->patches || pr->n_stamps <= 0 || !pr->stamps || 14 * pr->psf_K > 62)
        return CELESTE_ERR_INVALID_ARG;
    const int device = imgs->device;
    int st = select_device(device);
    if (st != CELESTE_OK) return st;

    celeste_ctx *c = new (std::nothrow) celeste_ctx();
    if (!c) return CELESTE_ERR_ALLOC;
    ArenaScope upload_arena(256u << 10);    // (declared after `c`: a failed creation destroys the context first, then the arena waits)
    c->device = device;
    c->imgs = imgs; imgs->refs.fetch_add(1);
    c->d_images = imgs->d_images;
    c->N = pr->n_images; c->S = pr->n_sources; c->K = pr->psf_K; c->NC = 14 * pr->psf_K;
    c->n_stamps = pr->n_stamps;
#define CTX_TRY(expr) do { int s__ = (expr); if (s__ != CELESTE_OK) { celeste_ctx_destroy(c); return s__; } } while (0)
    // (The copy stream carries the device-to-host copies of finished parts while the next part computes on `stream`.  HIP maps
    // streams onto a handful of hardware queues, and which queue a stream gets depends on how many streams the process has
    // created before: one context in four or so sweeps 2000 sources through the host-pointer entry in 2.0 - 2.4 ms instead of
    // 1.35 -- its copies do not overlap its kernels.  Creating the copy stream at high priority only moves the bad draw to other
    // contexts: tools/gpu_group_stream_lottery.py, profiles/r06_stream_lottery.txt.)
    if (stream_acquire(device, &c->stream) != hipSuccess || stream_acquire(device, &c->copy_stream) != hipSuccess) {
        celeste_ctx_destroy(c); return CELESTE_ERR_HIP;
    }

    // patches + explicit bitmaps (dense [s * N + n] table, or the sparse list sorted by (source, image)).
    // Every per-(source, image) table of the context is indexed by VISIT: the v-th non-empty patch in (source, image)
    // order (h_vis_off / h_vis_img / h_vis_src).  When every source appears in (nearly) every image the visit lists
    // simply enumerate all S x N pairs (empty patches included), visit id = s * N + n, and the kernels skip the lists.
    std::vector<uint8_t> pool;
    const bool sparse = pr->n_patch_entries > 0;
    if (sparse && (!pr->patch_source || !pr->patch_image)) { celeste_ctx_destroy(c); return CELESTE_ERR_INVALID_ARG; }
    const size_t n_entries = sparse ? (size_t)pr->n_patch_entries : (size_t)c->S * c->N;
    std::vector<int64_t> ent_key; std::vector<DevPatch> ent;     // the non-empty patches, by key s * N + n
    std::vector<int32_t> n_vis((size_t)c->S, 0);
    int64_t prev = -1;
    for (size_t k = 0; k < n_entries; ++k) {
        size_t q = k;
        if (sparse) {
            const int32_t s = pr->patch_source[k], n = pr->patch_image[k];
            if (s < 0 || s >= c->S || n < 0 || n >= c->N || (int64_t)s * c->N + n <= prev) {
                celeste_ctx_destroy(c); return CELESTE_ERR_INVALID_ARG;
            }
            prev = (int64_t)s * c->N + n;
            q = (size_t)prev;
        }
        const celeste_patch_t &p = pr->patches[k];
        const DevImage &im = imgs->h_images[q % c->N];
        DevPatch d; memset(&d, 0, sizeof d);
        d.off_h = p.off_h; d.off_w = p.off_w; d.H2 = p.H2 < 0 ? 0 : p.H2; d.W2 = p.W2 < 0 ? 0 : p.W2;
        if (d.H2 > 0 && d.W2 > 0 &&
            (p.off_h < 0 || p.off_w < 0 || p.off_h + d.H2 > im.H || p.off_w + d.W2 > im.W)) {
            celeste_ctx_destroy(c); return CELESTE_ERR_INVALID_ARG;
        }
        if (p.stamp < 0 || p.stamp >= pr->n_stamps || !p.psf) { celeste_ctx_destroy(c); return CELESTE_ERR_INVALID_ARG; }
        if (d.H2 * d.W2 <= 0) continue;                          // an empty patch is no visit
        d.stamp = p.stamp;
        d.bitmap_off = -1;
        if (p.bitmap) {
            d.bitmap_off = (int64_t)pool.size();
            pool.insert(pool.end(), p.bitmap, p.bitmap + (size_t)d.H2 * d.W2);
        }
        memcpy(d.J, p.wcs_jacobian, sizeof d.J);
        memcpy(d.wc, p.world_center, sizeof d.wc);
        memcpy(d.pc, p.pixel_center, sizeof d.pc);
        memcpy(d.psf, p.psf, sizeof(double) * 6 * c->K);
        ent_key.push_back((int64_t)q); ent.push_back(d);
        n_vis[q / c->N]++;
        if (d.H2 * d.W2 > c->max_npx) c->max_npx = d.H2 * d.W2;
    }
    for (int s = 0; s < c->S; ++s) c->M = std::max(c->M, n_vis[s]);
    // single-field problems (every source in every image, or nearly: at least 3/4 of the S x N pairs exist): list all
    // N images for every source and let the kernels map (target, j) -> image j directly
    c->dense = (int64_t)ent.size() * 4 >= (int64_t)c->S * c->N * 3 && !getenv("CELESTE_FORCE_VISIT_LISTS");   // (the variable exists for testing)
    if (c->dense) c->M = c->N;
    c->h_vis_off.assign((size_t)c->S + 1, 0);
    if (c->dense) {
        DevPatch empty; memset(&empty, 0, sizeof empty);
        empty.bitmap_off = -1;
        c->h_patches.assign((size_t)c->S * c->N, empty);
        for (size_t k = 0; k < ent.size(); ++k) c->h_patches[(size_t)ent_key[k]] = ent[k];
        for (int s = 0; s < c->S; ++s) {
            for (int n = 0; n < c->N; ++n) { c->h_vis_img.push_back(n); c->h_vis_src.push_back(s); }
            c->h_vis_off[s + 1] = (int32_t)c->h_vis_img.size();
        }
    } else {
        c->h_patches.swap(ent);
        for (size_t k = 0; k < ent_key.size(); ++k) {
            c->h_vis_src.push_back((int32_t)(ent_key[k] / c->N)); c->h_vis_img.push_back((int32_t)(ent_key[k] % c->N));
            c->h_vis_off[(size_t)(ent_key[k] / c->N) + 1] = (int32_t)(k + 1);
        }
        for (int s = 0; s < c->S; ++s) c->h_vis_off[s + 1] = std::max(c->h_vis_off[s + 1], c->h_vis_off[s]);
    }
    c->V = (int64_t)c->h_vis_img.size();
    if (c->V > 0x7fffffffll / std::max(c->NC, 1)) { celeste_ctx_destroy(c); return CELESTE_ERR_INVALID_ARG; }
    if (c->max_npx >= (1 << 22)) { celeste_ctx_destroy(c); return CELESTE_ERR_INVALID_ARG; }   // (divmod_small: pixel indices of a patch in 22 bits)
    CTX_TRY(dev_upload(&c->d_patches, c->h_patches.data(), c->h_patches.size()));
    CTX_TRY(dev_upload(&c->d_bitmaps, pool.data(), pool.size()));

    // conditioned + prefiltered star splines
    {
        // (on the device: spline_prefilter_kernel -- the raw stamps go up, the coefficient tables never exist on the host)
        double *d_stamps = nullptr;
        CTX_TRY(dev_upload(&d_stamps, pr->stamps, (size_t)c->n_stamps * CEL_STAMP * CEL_STAMP));
        int st_c = dev_upload<double>(&c->d_coefs, nullptr, (size_t)c->n_stamps * CEL_COEF * CEL_COEF);
        if (st_c == CELESTE_OK) st_c = dev_upload<float>(&c->d_coefs_f, nullptr, (size_t)c->n_stamps * CEL_COEF * CEL_COEF);
        if (st_c == CELESTE_OK) {
            hipLaunchKernelGGL(spline_prefilter_kernel, dim3((unsigned)c->n_stamps), dim3(64), 0, nullptr, d_stamps, c->d_coefs, c->d_coefs_f);
            if (hipGetLastError() != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) st_c = CELESTE_ERR_HIP;
        }
        (void)hipFree(d_stamps);
        CTX_TRY(st_c);
    }

    // neighbour CSR
    c->h_nbr_off.assign((size_t)c->S + 1, 0);
    if (pr->nbr_offsets) {
        for (int s = 0; s <= c->S; ++s) c->h_nbr_off[s] = pr->nbr_offsets[s];
        const int64_t tot = c->h_nbr_off[c->S];
        if (tot < 0 || (tot > 0 && !pr->nbr_index)) { celeste_ctx_destroy(c); return CELESTE_ERR_INVALID_ARG; }
        c->h_nbr_idx.assign(pr->nbr_index, pr->nbr_index + tot);
        for (int32_t v : c->h_nbr_idx) if (v < 0 || v >= c->S) { celeste_ctx_destroy(c); return CELESTE_ERR_INVALID_ARG; }
    }
    CTX_TRY(dev_upload(&c->d_nbr_off, c->h_nbr_off.data(), c->h_nbr_off.size()));
    CTX_TRY(dev_upload(&c->d_nbr_idx, c->h_nbr_idx.data(), c->h_nbr_idx.size()));

    // prior (+ inverse covariances, parameter independent)
    {
        PriorDev pd;
        pd.p = pr->prior ? *pr->prior : DEFAULT_PRIOR;
        for (int i = 0; i < 2; ++i) for (int d = 0; d < 8; ++d) inv4_logdet(pd.p.color_cov[i][d], pd.inv_cov[i][d], &pd.logdet[i][d]);
        CTX_TRY(dev_upload(&c->d_prior, &pd, 1));
    }
    {
        // the galaxy prototypes and the exponential's table are constants of the library: written once per device and process
        // (two synchronous symbol copies + a launch per context were 35 us of a per-source context's 320)
        static std::mutex consts_mu;
        static bool consts_ready[16] = {};
        std::lock_guard<std::mutex> lk(consts_mu);
        if (device < 0 || device >= 16 || !consts_ready[device]) {
            double eta[16], nu[16];
            galaxy_prototypes(eta, nu);
            if (hipMemcpyToSymbol(HIP_SYMBOL(c_eta), eta, sizeof eta) != hipSuccess ||
                hipMemcpyToSymbol(HIP_SYMBOL(c_nu), nu, sizeof nu) != hipSuccess) {
                celeste_ctx_destroy(c); return CELESTE_ERR_HIP;
            }
            hipLaunchKernelGGL(exp_table_kernel, dim3(1), dim3(64), 0, nullptr);
            if (hipStreamSynchronize(nullptr) != hipSuccess) { celeste_ctx_destroy(c); return CELESTE_ERR_HIP; }
            if (device >= 0 && device < 16) consts_ready[device] = true;
        }
    }
    CTX_TRY(dev_upload<SrcImg>(&c->d_srcimg, nullptr, (size_t)c->V));
    CTX_TRY(dev_upload<Comp>(&c->d_comps, nullptr, (size_t)c->V * c->NC));
    CTX_TRY(dev_upload<SrcGeo>(&c->d_geo, nullptr, (size_t)c->S));
    {
        std::vector<int64_t> voff(c->h_patches.size());
        int64_t tot = 0;
        for (size_t q = 0; q < c->h_patches.size(); ++q) { voff[q] = tot; tot += (int64_t)c->h_patches[q].H2 * c->h_patches[q].W2; }
        CTX_TRY(dev_upload(&c->d_val_off, voff.data(), voff.size()));
        CTX_TRY(dev_upload<double2>(&c->d_val, nullptr, (size_t)tot));
        CTX_TRY(dev_upload<int32_t>(&c->d_needed, nullptr, (size_t)c->S));
        if (hipMemset(c->d_needed, 0, (size_t)c->S * sizeof(int32_t)) != hipSuccess) { celeste_ctx_destroy(c); return CELESTE_ERR_HIP; }
    }

    CTX_TRY(dev_upload(&c->d_vis_off, c->h_vis_off.data(), c->h_vis_off.size()));
    CTX_TRY(dev_upload(&c->d_vis_img, c->h_vis_img.data(), c->h_vis_img.size()));
    CTX_TRY(dev_upload(&c->d_vis_src, c->h_vis_src.data(), c->h_vis_src.size()));
    CTX_TRY(dev_upload<double>(&c->d_lg_sum, nullptr, (size_t)c->V));
    if (c->V > 0)
        hipLaunchKernelGGL(patch_lgamma_kernel, dim3((unsigned)c->V), dim3(256), 0, nullptr, c->d_images, c->d_patches,
                           c->d_vis_img, c->d_bitmaps, c->d_lg_sum);
    const char *env_chunk = getenv("CELESTE_CHUNK_PX");
    if (env_chunk && atoi(env_chunk) >= 64) c->chunk_px = (atoi(env_chunk) + 63) / 64 * 64;
    c->CH = c->max_npx > 0 ? (c->max_npx + c->chunk_px - 1) / c->chunk_px : 1;
    c->h_src_chunks.assign((size_t)c->S, 0);
    for (int s = 0; s < c->S; ++s) {
        for (int v = c->h_vis_off[s]; v < c->h_vis_off[s + 1]; ++v) {
            const DevPatch &q = c->h_patches[v];
            c->h_src_chunks[s] += (q.H2 * q.W2 + c->chunk_px - 1) / c->chunk_px;
        }
        c->max_src_chunks = std::max(c->max_src_chunks, c->h_src_chunks[s]);
    }
    // visit of (source, image), -1: none
    auto hv = [c](int s, int n) -> int64_t {
        if (c->dense) return (int64_t)s * c->N + n;
        const int32_t *b = c->h_vis_img.data() + c->h_vis_off[s], *e = c->h_vis_img.data() + c->h_vis_off[s + 1];
        const int32_t *it = std::lower_bound(b, e, n);
        return it != e && *it == n ? (int64_t)(it - c->h_vis_img.data()) : -1;
    };
    if (!c->dense) {
        // the visits of every source's neighbours, image by image: row j (= visit vis_off[s] + j) of source s holds the
        // visit of each of its K neighbours in that visit's image (-1: the neighbour is not in it), at nv_base[s] + j K
        std::vector<int64_t> base((size_t)c->S + 1, 0);
        for (int s = 0; s < c->S; ++s)
            base[s + 1] = base[s] + (int64_t)(c->h_vis_off[s + 1] - c->h_vis_off[s]) * (c->h_nbr_off[s + 1] - c->h_nbr_off[s]);
        std::vector<int32_t> nv((size_t)base[c->S]);
        for (int s = 0; s < c->S; ++s) {
            const int64_t K = c->h_nbr_off[s + 1] - c->h_nbr_off[s];
            for (int v = c->h_vis_off[s]; v < c->h_vis_off[s + 1]; ++v)
                for (int64_t q = 0; q < K; ++q)
                    nv[base[s] + (int64_t)(v - c->h_vis_off[s]) * K + q] = (int32_t)hv(c->h_nbr_idx[c->h_nbr_off[s] + q], c->h_vis_img[v]);
        }
        CTX_TRY(dev_upload(&c->d_nv_base, base.data(), base.size()));
        CTX_TRY(dev_upload(&c->d_nbr_vis, nv.data(), nv.size()));
    }
    {
        std::vector<int32_t> sorted(c->h_src_chunks);
        std::sort(sorted.begin(), sorted.end(), [](int32_t a, int32_t b) { return a > b; });
        c->h_top_chunks.assign((size_t)c->S + 1, 0);
        for (int s = 0; s < c->S; ++s) c->h_top_chunks[s + 1] = c->h_top_chunks[s] + sorted[s];
    }
    CTX_TRY(dev_upload<int32_t>(&c->d_work_total, nullptr, 1));
    {
        // work items of value_kernel: (link, image, chunk) with a non-empty overlap rectangle, longest first (by the
        // number of 64-pixel iterations the chunk takes) so that the kernel's tail is made of its shortest workgroups
        if (c->N > 0xffff) { celeste_ctx_destroy(c); return CELESTE_ERR_INVALID_ARG; }
        const int n_cls = c->chunk_px / 64;
        std::vector<std::vector<std::pair<int32_t, int32_t>>> by_len(n_cls);   // [n_cls - iterations]
        for (int s = 0; s < c->S; ++s)
            for (int64_t q = c->h_nbr_off[s]; q < c->h_nbr_off[s + 1]; ++q) {
                const int s2 = c->h_nbr_idx[q];
                for (int va = c->h_vis_off[s]; va < c->h_vis_off[s + 1]; ++va) {
                    const int n = c->h_vis_img[va];
                    const int64_t vb = hv(s2, n);
                    if (vb < 0) continue;
                    const DevPatch &a = c->h_patches[va], &b = c->h_patches[vb];
                    const int rh = std::min(a.off_h + a.H2, b.off_h + b.H2) - std::max(a.off_h, b.off_h);
                    const int rw = std::min(a.off_w + a.W2, b.off_w + b.W2 - 1) - std::max(a.off_w, b.off_w);
                    if (rh <= 0 || rw <= 0) continue;
                    const int npx = rh * rw, nch = (npx + c->chunk_px - 1) / c->chunk_px;
                    for (int ch = 0; ch < nch; ++ch) {
                        const int px = std::min(c->chunk_px, npx - ch * c->chunk_px);
                        by_len[n_cls - (px + 63) / 64].push_back({(int32_t)q, (int32_t)(n | (ch << 16))});
                    }
                }
            }
        std::vector<int32_t> link_src(c->h_nbr_idx.size());
        for (int s = 0; s < c->S; ++s)
            for (int64_t q = c->h_nbr_off[s]; q < c->h_nbr_off[s + 1]; ++q) link_src[q] = s;
        std::vector<int4> desc;
        for (auto &v : by_len)
            for (auto &e : v) {
                const int t = link_src[e.first], s2 = c->h_nbr_idx[e.first], n = e.second & 0xffff, ch = e.second >> 16;
                desc.push_back(make_int4((int)hv(s2, n), (int)hv(t, n), ch, t));
            }
        c->n_value_items = (int64_t)desc.size();
        CTX_TRY(dev_upload(&c->d_value_items, desc.data(), desc.size()));
        // the same items grouped by target (joint dataflow launch: an entry renders its own source's neighbours)
        c->h_vitem_off.assign((size_t)c->S + 1, 0);
        for (auto &d : desc) c->h_vitem_off[(size_t)d.w + 1]++;
        for (int s = 0; s < c->S; ++s) c->h_vitem_off[s + 1] += c->h_vitem_off[s];
        {
            std::vector<int4> by_src(desc.size());
            std::vector<int32_t> fill(c->h_vitem_off.begin(), c->h_vitem_off.end() - 1);
            for (auto &d : desc) by_src[(size_t)fill[d.w]++] = d;
            CTX_TRY(dev_upload(&c->d_vitems_src, by_src.data(), by_src.size()));
            CTX_TRY(dev_upload(&c->d_vitem_off, c->h_vitem_off.data(), c->h_vitem_off.size()));
        }
        CTX_TRY(dev_upload<int32_t>(&c->d_prep_mark, nullptr, (size_t)c->S));
        if (hipMemset(c->d_prep_mark, 0, (size_t)c->S * sizeof(int32_t)) != hipSuccess) { celeste_ctx_destroy(c); return CELESTE_ERR_HIP; }
    }
    {
        c->h_tile_off.resize(c->h_patches.size());
        int64_t tot = 0;
        for (size_t q = 0; q < c->h_patches.size(); ++q) {
            c->h_tile_off[q] = tot;
            tot += ((int64_t)c->h_patches[q].H2 * c->h_patches[q].W2 + 63) / 64;
        }
        c->n_tiles = tot;
        if (const char *e = getenv("CELESTE_SUM_TILES")) if (atoi(e) > 0) c->sum_tiles = atoi(e);
        c->RCH = std::max(1, ((c->max_npx + 63) / 64 + c->sum_tiles - 1) / c->sum_tiles);
    }
    // (the timing events and the part events of the host-pointer sweep are created on first use: celeste_ctx_enable_timing,
    // celeste_elbo_eval_batch -- a per-source context that serves a few one-target calls never needs them)
    // the constant tables above were written on the NULL stream; the context's own streams do not wait for it
    if (hipStreamSynchronize(nullptr) != hipSuccess) { celeste_ctx_destroy(c); return CELESTE_ERR_HIP; }
#undef CTX_TRY
    *out = c;
    return CELESTE_OK;
} ABI_CATCH

extern "C" void celeste_ctx_destroy(celeste_ctx_t *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    void *ptrs[] = {c->d_patches, c->d_coefs, c->d_coefs_f, c->d_bitmaps, c->d_nbr_off, c->d_nbr_idx, c->d_prior,
                    c->d_srcimg, c->d_comps, c->d_geo, c->d_val_off, c->d_val, c->d_needed, c->d_vis_off, c->d_vis_img, c->d_vis_src, c->d_value_items, c->d_vitems_src, c->d_vitem_off, c->d_prep_mark, c->d_rec_off, c->d_lg_sum, c->d_nv_base, c->d_nbr_vis, c->d_items, c->d_work, c->d_work_blk, c->d_work_total, c->d_tile_off, c->d_rec, c->d_acc_split, c->d_acc, c->d_vp, c->d_targets, c->d_v, c->d_d, c->d_h,
                    c->d_cnt, c->d_status};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    {
        auto &o = c->opt;
        void *optr[] = {o.d_vp, o.d_v, o.d_d, o.d_h, o.d_H, o.d_T, o.d_S, o.d_pos, o.d_targets, o.d_act[0], o.d_act[1], o.d_evt[0], o.d_evt[1],
                        o.d_count, o.d_st, o.d_state};
        for (void *q : optr) if (q) (void)hipFree(q);
        void *hptr[] = {o.h_vp, o.h_state, o.h_count};
        for (void *q : hptr) if (q) staging_free(q);
        for (int k = 0; k < celeste_ctx::OptBuffers::RING; ++k) if (o.ev[k]) (void)hipEventDestroy(o.ev[k]);
        auto &f = c->fused;
        void *fptr[] = {f.d_chunk_desc, f.d_tgt_rec, f.d_q_items, f.d_q_ctl, f.d_arrivals, f.d_saved};
        for (void *q : fptr) if (q) (void)hipFree(q);
        if (f.h_ctl) staging_free(f.h_ctl);
    }
    for (auto &sc : c->scratch) if (sc.p) (void)hipFree(sc.p);
    for (int i = 0; i < 5; ++i) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    for (int i = 0; i < celeste_ctx::MAX_PARTS; ++i) {
        if (c->part_done[i]) (void)hipEventDestroy(c->part_done[i]);
        if (c->part_copied[i]) (void)hipEventDestroy(c->part_copied[i]);
    }
    void *pins[] = {c->p_vp, c->p_targets, c->p_status, c->p_v, c->p_d, c->p_h, c->p_cnt, c->p_small_in, c->p_small_out};
    if (c->d_small_in) (void)hipFree(c->d_small_in);
    if (c->d_small_out) (void)hipFree(c->d_small_out);
    for (void *q : pins) if (q) staging_free(q);
    stream_retire(c->device, c->stream);
    stream_retire(c->device, c->copy_stream);
    images_release(c->imgs);
    delete c;
}

// slot `k` of the context's scratch arena, at least `bytes` large (the stream is drained before a buffer moves)
template <class T>
static hipError_t scratch_get(celeste_ctx_t *c, int k, size_t bytes, T **out) {
    auto &s = c->scratch[k];
    if (bytes > s.cap) {
        hipError_t e = s.p ? hipStreamSynchronize(c->stream) : hipSuccess;     // (only a block in use has to be waited for)
        if (e != hipSuccess) return e;
        if (s.p) { (void)hipFree(s.p); s.p = nullptr; s.cap = 0; }
        e = hipMalloc(&s.p, std::max<size_t>(bytes, 256));
        if (e != hipSuccess) return e;
        s.cap = std::max<size_t>(bytes, 256);
    }
    *out = (T *)s.p;
    return hipSuccess;
}

static int launch_eval(celeste_ctx_t *c, const double *d_vp, int32_t n_targets, const int32_t *d_targets,
                       uint32_t flags, double *d_v, double *d_d, double *d_h, int64_t *d_counters, int32_t *d_status,
                       void *stream_, bool render_neighbors, const int32_t *d_active_rank = nullptr,
                       int64_t n_chunks = -1, bool tables_current = false, const int32_t *d_live = nullptr,
                       bool prep_all = false, bool render_only = false, const int32_t *d_h_pos = nullptr);

extern "C" int celeste_elbo_eval_batch_device(celeste_ctx_t *c, const double *d_vp, int32_t n_targets,
                                              const int32_t *d_targets, uint32_t flags, double *d_v, double *d_d,
                                              double *d_h, int64_t *d_counters, int32_t *d_status, void *stream_) try {
    return launch_eval(c, d_vp, n_targets, d_targets, flags, d_v, d_d, d_h, d_counters, d_status, stream_, true);
} ABI_CATCH

// the events of the first n parts of a host-pointer sweep (created on first use; the calling thread's device is the context's)
static int ensure_part_events(celeste_ctx_t *c, int n) {
    for (int k = 0; k < n && k < celeste_ctx::MAX_PARTS; ++k) {
        if (!c->part_done[k]) HIP_TRY(hipEventCreateWithFlags(&c->part_done[k], hipEventDisableTiming));
        if (!c->part_copied[k]) HIP_TRY(hipEventCreateWithFlags(&c->part_copied[k], hipEventDisableTiming));
    }
    return CELESTE_OK;
}

// the context's tables, as the fused kernels take them
static void fused_args_tables(celeste_ctx_t *c, FusedArgs &A) {
    A.images = c->d_images; A.patches = c->d_patches; A.coefs = c->d_coefs; A.bitmaps = c->d_bitmaps;
    A.nbr_off = c->d_nbr_off; A.nbr_idx = c->d_nbr_idx; A.val_off = c->d_val_off; A.val = c->d_val;
    A.nv_base = c->d_nv_base; A.nbr_vis = c->d_nbr_vis; A.items = c->dense ? nullptr : c->d_items; A.geo = c->d_geo;
    A.prior = c->d_prior; A.vis_off = c->d_vis_off; A.vis_img = c->d_vis_img; A.lg_sum = c->d_lg_sum; A.rec_off = c->d_rec_off;
    A.N = c->N; A.NC = c->NC; A.K = c->K; A.M = c->M; A.CH = c->CH; A.chunk_px = c->chunk_px;
    A.acc = c->d_acc;
    A.st = nullptr; A.Hstate = nullptr; A.Tstate = nullptr; A.Spec = nullptr; A.q_items = nullptr; A.q_ctl = nullptr; A.q_cap = 0; A.timeout_ticks = 0;
    A.j_R = 0; A.j_gshift = 0; A.j_dep = nullptr; A.j_succ_off = nullptr; A.j_succ = nullptr; A.j_vitem_off = nullptr;
    A.j_vitems = nullptr; A.j_render_arr = nullptr; A.j_saved = nullptr; A.j_pos = nullptr; A.j_srcimg = nullptr;
    A.j_comps = nullptr; A.j_geo = nullptr;
}

// per-target / per-record buffers of the fused kernels at capacity >= (n, rec)
static int fused_buffers(celeste_ctx_t *c, size_t n, size_t rec, hipStream_t stream) {
    auto &fb = c->fused;
    if (n > fb.cap_t) {
        if (fb.d_tgt_rec || fb.d_arrivals) HIP_TRY(hipStreamSynchronize(stream));   // (a buffer in use is about to be freed; nothing to wait for the first time)
        void **ps[] = {(void **)&fb.d_tgt_rec, (void **)&fb.d_arrivals};
        for (void **q : ps) if (*q) { (void)hipFree(*q); *q = nullptr; }
        fb.cap_t = 0;
        HIP_TRY(hipMalloc((void **)&fb.d_tgt_rec, n * sizeof(int2)));
        HIP_TRY(hipMalloc((void **)&fb.d_arrivals, n * sizeof(int32_t)));
        fb.cap_t = n;
        fb.arrivals_dirty = true;
    }
    if (rec > fb.cap_rec) {
        if (fb.d_chunk_desc) { HIP_TRY(hipStreamSynchronize(stream)); (void)hipFree(fb.d_chunk_desc); fb.d_chunk_desc = nullptr; }
        fb.cap_rec = 0;
        HIP_TRY(hipMalloc((void **)&fb.d_chunk_desc, std::max<size_t>(rec, 1) * 2 * sizeof(int4)));
        fb.cap_rec = rec;
    }
    return CELESTE_OK;
}

#define VALUE_WIDE_MAX 768     // batches up to this size render the neighbours' light with four wavefronts per item
// render_neighbors = false keeps the neighbours' pre-rendered light of an earlier call (frozen neighbours
// during an optimisation, ParallelRun.jl:474-488)
static int launch_eval(celeste_ctx_t *c, const double *d_vp, int32_t n_targets, const int32_t *d_targets,
                       uint32_t flags, double *d_v, double *d_d, double *d_h, int64_t *d_counters, int32_t *d_status,
                       void *stream_, bool render_neighbors, const int32_t *d_active_rank, int64_t n_chunks,
                       bool tables_current, const int32_t *d_live, bool prep_all, bool render_only, const int32_t *d_h_pos) {
    // d_h_pos (device, optional): target i's Hessian goes to slot d_h_pos[i] of d_h instead of slot i (celeste_group_*: a
    // member's shard written to its places in the caller's array)
    // render_only: the batch's bookkeeping (SrcGeo, visit items, record offsets, marks), the per-image tables of the
    // targets and their neighbours and the neighbours' pre-rendered light -- everything the optimiser needs before its
    // first evaluation -- and no evaluation
    // prep_all: the per-(source, image) tables of EVERY source are filled (the first part of a host batch does it for
    // the parts that follow, which pass tables_current); else only those of the targets and their neighbours
    // d_live (device, optional): the number of leading entries of d_targets that are live; n_targets is then an upper
    // bound known to the host (the optimiser loop runs ahead of the device)
    // tables_current: the per-(source, image) tables were filled from this very vp by an earlier launch of the same
    // call (parts of one host batch): only the neighbours of this part's targets are rendered
    if (!c || !d_vp || !d_targets || !d_v || !d_status || n_targets < 0) return CELESTE_ERR_INVALID_ARG;
    if ((flags & CELESTE_FLAG_HESS) && !d_h) return CELESTE_ERR_INVALID_ARG;
    if ((flags & (CELESTE_FLAG_GRAD | CELESTE_FLAG_HESS)) && !d_d) return CELESTE_ERR_INVALID_ARG;
    if (n_targets == 0) return CELESTE_OK;
    hipStream_t stream = (hipStream_t)stream_;
    HIP_TRY(hipSetDevice(c->device));
    // Pixel chunks of THIS launch.  Single precision: 512 pixels, i.e. four 128-pixel trips of pixel_iter_px2 per record
    // instead of two -- half the records to zero, fold, store and lift (measured on config 5: 5.31 -> 5.20 ms; the fp64 kernel
    // loses 3-5 % with 512, so the context's own chunk stays 256).  Everything that depends on the chunk size is per launch
    // (work list, record offsets, pixel and lift kernels); the host's chunk counts (256-pixel chunks) remain upper bounds.
    const bool big_chunks = (flags & CELESTE_FLAG_FP32) && !(flags & CELESTE_FLAG_SPLIT) && !d_active_rank && c->chunk_px == 256 &&
                            !getenv("CELESTE_FP32_CHUNK_256");
    const int chunk_px = big_chunks ? 512 : c->chunk_px;
    const int CH = big_chunks ? std::max(1, (c->max_npx + chunk_px - 1) / chunk_px) : c->CH;
    // one 68-double record per chunk that exists (rec_off = prefix sum of the visits' chunk counts): exactly n_chunks
    // when the caller knows its targets on the host, else at most n_targets x the chunk count of the richest source
    const size_t rec_cap = n_chunks >= 0 ? (size_t)n_chunks
                                         : std::min((size_t)n_targets * c->max_src_chunks, (size_t)n_targets * c->M * c->CH);
    const size_t need = std::max<size_t>(rec_cap, 1) * ACC_N;
    if ((size_t)n_targets * c->M > c->rec_off_cap) {
        if (c->d_rec_off) { HIP_TRY(hipStreamSynchronize(stream)); HIP_TRY(hipFree(c->d_rec_off)); c->d_rec_off = nullptr; }
        HIP_TRY(hipMalloc((void **)&c->d_rec_off, (size_t)n_targets * c->M * sizeof(int32_t)));
        c->rec_off_cap = (size_t)n_targets * c->M;
    }
    if (need > c->acc_cap) {
        if (c->d_acc) { HIP_TRY(hipStreamSynchronize(stream)); HIP_TRY(hipFree(c->d_acc)); c->d_acc = nullptr; }
        HIP_TRY(hipMalloc((void **)&c->d_acc, need * sizeof(double)));
        c->acc_cap = need;
    }
    if (!c->dense && (size_t)n_targets * c->M > c->items_cap) {
        if (c->d_items) { HIP_TRY(hipStreamSynchronize(stream)); HIP_TRY(hipFree(c->d_items)); c->d_items = nullptr; }
        HIP_TRY(hipMalloc((void **)&c->d_items, (size_t)n_targets * c->M * sizeof(int2)));
        c->items_cap = (size_t)n_targets * c->M;
    }
    // work list: n_chunks = number of chunks of the batch when the caller knows its targets on the host, else bounded
    // by the largest source (workgroups past the device-side total exit on one scalar load)
    const int n_visits = n_targets * c->M;
    const int n_wblk = (n_visits + WORK_NT - 1) / WORK_NT;
    // n_chunks: exact when the caller knows its targets on the host; else the chunks of the n_targets chunk-richest
    // sources (a bound for distinct targets; with repeated targets the list can be longer: the buffer is sized for
    // that, the pixel kernel's stride loop covers it)
    const size_t work_need = (size_t)(n_chunks >= 0 ? n_chunks : (int64_t)n_targets * c->max_src_chunks);
    const size_t grid_need = (size_t)(n_chunks >= 0 ? n_chunks
                                      : n_targets <= c->S ? c->h_top_chunks[n_targets] : (int64_t)n_targets * c->max_src_chunks);
    if ((size_t)n_targets * c->M * c->CH > 0x7fffffffull) return CELESTE_ERR_INVALID_ARG;
    if (work_need > c->work_cap) {
        if (c->d_work) { HIP_TRY(hipStreamSynchronize(stream)); HIP_TRY(hipFree(c->d_work)); c->d_work = nullptr; }
        HIP_TRY(hipMalloc((void **)&c->d_work, std::max<size_t>(work_need, 1) * sizeof(int32_t)));
        c->work_cap = work_need;
    }
    // a workgroup handles a group of up to G consecutive chunks of a patch (results do not depend on G, see
    // visit_chunks): 4 for large sweeps, 1 for small batches whose critical path is one patch
    // (measured, round 5: the 2000-target sweep of the bench field 0.499 ms with 2, 0.503 with 1, 0.502 with 3, 0.512 with 4;
    // config 5's 30 000 targets 5.31 ms with 4, 5.45 with 2, 5.30 with 8; a 1000-target shard 0.287 ms with 2, 0.315 with 4)
    int G = n_targets >= 8192 ? 4 : n_targets >= 768 ? 2 : 1;
    if (const char *e = getenv("CELESTE_CHUNK_GROUP")) if (atoi(e) >= 1 && atoi(e) <= 16) G = atoi(e);
    // small Hessian-mode fp64 batches: eval_fused_kernel instead of pixel_kernel + lift_kernel (below)
    bool eval_fused = !render_only && (flags & CELESTE_FLAG_HESS) && !(flags & (CELESTE_FLAG_FP32 | CELESTE_FLAG_SPLIT)) &&
                      !d_active_rank && !d_live && !d_h_pos && d_d && d_h && n_targets <= EVAL_FUSED_MAX;
    if (const char *e = getenv("CELESTE_EVAL_FUSED")) {
        if (atoi(e) == 0) eval_fused = false;
        else if (atoi(e) == 1) eval_fused = !render_only && (flags & CELESTE_FLAG_HESS) && !(flags & (CELESTE_FLAG_FP32 | CELESTE_FLAG_SPLIT)) &&
                                          !d_active_rank && !d_live && !d_h_pos && d_d && d_h;
    }
    if (eval_fused) G = 1;        // one record per work item: the work-list total is the batch's number of records
    const int n_classes = WORK_CLASSES;   // work-list classes: full groups, then the patches' last groups by length
    if ((size_t)n_wblk * (n_classes + 1) > c->work_blk_cap) {   // + the row of chunk counts (rec_off)
        if (c->d_work_blk) { HIP_TRY(hipStreamSynchronize(stream)); HIP_TRY(hipFree(c->d_work_blk)); c->d_work_blk = nullptr; }
        HIP_TRY(hipMalloc((void **)&c->d_work_blk, (size_t)n_wblk * (n_classes + 1) * sizeof(int32_t)));
        c->work_blk_cap = (size_t)n_wblk * (n_classes + 1);
    }
    const bool derivs = (flags & (CELESTE_FLAG_GRAD | CELESTE_FLAG_HESS)) != 0;
    const bool split = (flags & CELESTE_FLAG_SPLIT) != 0;
    if (split) {
        if (!(flags & CELESTE_FLAG_HESS)) return CELESTE_ERR_INVALID_ARG;
        if (!c->d_rec) {
            HIP_TRY(hipStreamSynchronize(stream));
            if (dev_upload(&c->d_tile_off, c->h_tile_off.data(), c->h_tile_off.size()) != CELESTE_OK) return CELESTE_ERR_HIP;
            if (hipMalloc((void **)&c->d_rec, (size_t)std::max<int64_t>(c->n_tiles, 1) * ACC_N * 64 * sizeof(double)) != hipSuccess)
                return CELESTE_ERR_ALLOC;
        }
        const size_t need_s = (size_t)n_targets * c->M * c->RCH * ACC_N;
        if (need_s > c->acc_split_cap) {
            if (c->d_acc_split) { HIP_TRY(hipStreamSynchronize(stream)); HIP_TRY(hipFree(c->d_acc_split)); c->d_acc_split = nullptr; }
            HIP_TRY(hipMalloc((void **)&c->d_acc_split, need_s * sizeof(double)));
            c->acc_split_cap = need_s;
        }
    }
    if (c->timing) HIP_TRY(hipEventRecord(c->ev[0], stream));
    if (render_neighbors && ++c->stamp == 0x7fffffff) {   // (the marks are batch stamps; 0 = never a target)
        HIP_TRY(hipMemsetAsync(c->d_needed, 0, (size_t)c->S * sizeof(int32_t), stream));
        c->stamp = 1;
    }
    // SrcGeo: of every source when the neighbours are (re)rendered, else of the targets only (setup_thread)
    const int geo_S = render_neighbors ? c->S : -1;
    const size_t setup_threads = std::max<size_t>(render_neighbors ? (size_t)c->S : 0, (size_t)n_targets * (c->dense ? 1 : c->M));
    // the sources whose tables this launch fills are marked by the setup kernel (targets + neighbours)
    int32_t *const prep_mark = render_neighbors && !tables_current && !prep_all ? c->d_prep_mark : nullptr;
    bool prep_fused = false;
    bool records_described = false;   // eval_fused: block 0 of the one-launch list has written chunk_desc / tgt_rec (no fused_setup_kernel)
    if (eval_fused) { int stb = fused_buffers(c, (size_t)n_targets, rec_cap, stream); if (stb != CELESTE_OK) return stb; }
    if (n_visits <= WORK1_MAX_VISITS && !getenv("CELESTE_PARALLEL_WORKLIST")) {
        const unsigned setup_blocks = (unsigned)((setup_threads + WORK1_SETUP - 1) / WORK1_SETUP);
        // neighbours frozen: the targets' tables are filled by extra blocks of this launch (one launch less per iteration)
        prep_fused = !render_neighbors && !getenv("CELESTE_NO_FUSED_PREP");
        // neighbours (re)rendered: every visit's tables by this launch as well when the context has few of them (a one-field
        // context: 10 000 visits, 15 us of work that hides under the list builder's block) -- a small batch's chain is one
        // launch shorter
        // (a handful of targets touch a handful of sources: there the marked tables of prep_kernel are the cheaper route)
        // (... unless the context is tiny -- a per-source context, a source and its neighbours: ten visits -- where filling
        // every table inside this launch beats a launch of its own whatever the batch: a one-target call 71 -> 67 us)
        const bool prep_all_here = render_neighbors && !tables_current && c->V > 0 && c->V <= WORK1_PREP_ALL_MAX &&
                                   (n_targets >= WORK1_PREP_ALL_MIN_TARGETS || c->V <= WORK1_PREP_ALL_TINY) && !getenv("CELESTE_NO_FUSED_PREP");
        const int prep_n = prep_all_here ? (int)c->V : n_visits;
        if (prep_all_here) prep_fused = true;
        const unsigned prep_blocks = prep_fused ? (unsigned)((prep_n + WORK1_NT / 64 - 1) / (WORK1_NT / 64)) : 0u;
        const bool records_described_here = eval_fused && c->dense && !d_live && !getenv("CELESTE_NO_FUSED_DESCRIBE");
        hipLaunchKernelGGL(setup_worklist_kernel, dim3(1 + setup_blocks + prep_blocks), dim3(WORK1_NT),
                           0, stream, d_vp, geo_S, c->d_geo, d_targets, n_targets, c->d_vis_off, c->d_vis_img, c->M,
                           c->dense ? nullptr : c->d_items, render_neighbors ? c->d_needed : nullptr, c->stamp, c->d_patches,
                           c->N, CH, chunk_px, G, (int)c->dense, c->d_work, c->d_work_total, d_live, prep_mark,
                           c->d_nbr_off, c->d_nbr_idx, c->d_rec_off, (int)setup_blocks, c->d_images, c->K, c->d_srcimg,
                           c->d_comps, prep_all_here ? (int)c->V : 0, c->d_vis_src,
                           records_described_here ? c->fused.d_chunk_desc : nullptr, records_described_here ? c->fused.d_tgt_rec : nullptr,
                           c->chunk_px);
        records_described = records_described_here;
    } else {
        // (the work-list kernels depend on nothing setup / prep / value produce; built on a second stream beside them and
        // joined in front of the pixel kernel they measured no faster -- 0.6519 ms per sweep of the bench field either way --
        // and a stream more per context is not free: round 5, removed)
        hipLaunchKernelGGL(setup_kernel, dim3((unsigned)((setup_threads + 63) / 64)), dim3(64), 0, stream, d_vp, geo_S,
                           c->d_geo, d_targets, n_targets, c->d_vis_off, c->d_vis_img, c->M, c->dense ? nullptr : c->d_items,
                           render_neighbors ? c->d_needed : nullptr, c->stamp, prep_mark, c->d_nbr_off, c->d_nbr_idx, d_live);
        hipLaunchKernelGGL(work_count_kernel, dim3(n_wblk), dim3(WORK_NT), 0, stream, d_targets, n_visits, c->d_patches,
                           c->d_vis_off, c->d_vis_img, c->N, c->M, chunk_px, G, (int)c->dense, c->d_work_blk, d_live);
        hipLaunchKernelGGL(work_scan_kernel, dim3(1), dim3(1024), 0, stream, c->d_work_blk, n_wblk * (n_classes + 1),
                           c->d_work_total, n_wblk * n_classes);
        hipLaunchKernelGGL(work_fill_kernel, dim3(n_wblk), dim3(WORK_NT), 0, stream, d_targets, n_visits, c->d_patches,
                           c->d_vis_off, c->d_vis_img, c->N, c->M, CH, chunk_px, G, (int)c->dense, c->d_work_blk, c->d_work, d_live,
                           c->d_rec_off);
    }
    // per-(source, image) constants: of every source when the neighbours are (re)rendered, else of the targets only
    if (render_neighbors) {
        if (c->V > 0 && !tables_current && !prep_fused)
            hipLaunchKernelGGL(prep_kernel, dim3((unsigned)c->V), dim3(64), 0, stream, d_vp, c->d_images, c->d_patches,
                               c->d_vis_src, c->d_vis_img, c->N, c->K, c->d_srcimg, c->d_comps, nullptr, c->d_vis_off, c->M,
                               (int)c->dense, nullptr, prep_mark, c->stamp);
    } else if (!prep_fused) {
        hipLaunchKernelGGL(prep_kernel, dim3((unsigned)std::max(n_visits, 1)), dim3(64), 0, stream, d_vp, c->d_images,
                           c->d_patches, c->d_vis_src, c->d_vis_img, c->N, c->K, c->d_srcimg, c->d_comps, d_targets,
                           c->d_vis_off, c->M, (int)c->dense, d_live, nullptr, 0);
    }
    if (render_neighbors) {
    if (c->n_value_items > 0)
    {
        // (single-precision mode: the neighbours' densities in fp32 too, their moments in fp64)
        // small batches leave the chip mostly idle: an item's four 64-pixel iterations then run side by side on four
        // wavefronts (same values; chunk_px = 256 is what the items were cut for)
        const bool wide_items = n_targets <= VALUE_WIDE_MAX && c->chunk_px == 256 && !getenv("CELESTE_NO_WIDE_VALUE");
#define LAUNCH_VALUE(R, WAVES)                                                                                              \
        hipLaunchKernelGGL((value_kernel<R, WAVES>), dim3((unsigned)c->n_value_items), dim3(64 * WAVES), 0, stream,       \
                           c->d_patches, c->d_coefs, c->d_srcimg, c->d_comps, c->d_needed, c->stamp, c->d_val_off,          \
                           c->d_value_items, c->NC, c->chunk_px, c->d_val, c->d_coefs_f)
        // (single precision: value_pixels_f2 for every batch size -- two wavefronts x 128 pixels when wide -- so that a
        // target's fp32 result does not depend on the size of the batch it is in)
        if (flags & CELESTE_FLAG_FP32) { if (wide_items) LAUNCH_VALUE(float, 2); else LAUNCH_VALUE(float, 1); }
        else { if (wide_items) LAUNCH_VALUE(double, 4); else LAUNCH_VALUE(double, 1); }
#undef LAUNCH_VALUE
    }
    }
    if (render_only) { HIP_TRY(hipGetLastError()); return CELESTE_OK; }
    if (c->timing) HIP_TRY(hipEventRecord(c->ev[1], stream));
    if (eval_fused) {
        // small batch: one workgroup per chunk record (four wavefronts, one 64-pixel iteration each), the lift by the
        // workgroup that completes a target -- eval_fused_kernel; same records, same results as pixel_kernel + lift_kernel
        auto &fb = c->fused;
        if (fb.arrivals_dirty) {
            HIP_TRY(hipMemsetAsync(fb.d_arrivals, 0, fb.cap_t * sizeof(int32_t), stream));
            fb.arrivals_dirty = false;
        }
        if (!records_described)
            hipLaunchKernelGGL(fused_setup_kernel, dim3((unsigned)((n_targets + 63) / 64)), dim3(64), 0, stream, d_targets, n_targets,
                               c->d_patches, c->d_vis_off, c->dense ? nullptr : c->d_items, c->N, c->M, c->chunk_px, c->d_rec_off,
                               fb.d_chunk_desc, fb.d_tgt_rec, (int32_t *)nullptr, (int32_t *)nullptr);
        FusedArgs A;
        fused_args_tables(c, A);
        A.targets = d_targets; A.n_targets = n_targets; A.vp = const_cast<double *>(d_vp);
        A.chunk_desc = fb.d_chunk_desc; A.tgt_rec = fb.d_tgt_rec; A.arrivals = fb.d_arrivals; A.flags = flags;
        memset(&A.op, 0, sizeof A.op);
        // no stride loop in this kernel: the grid must cover every record.  h_top_chunks bounds DISTINCT targets only, and
        // the device-pointer entry allows repeats -- there the bound is n_targets x the chunk-richest source
        const unsigned rec_bound = (unsigned)std::max<size_t>(n_chunks >= 0 ? grid_need : work_need, 1);
        hipLaunchKernelGGL(eval_fused_kernel, dim3(rec_bound + (unsigned)n_targets), dim3(FUSED_NT), 0, stream, A, c->d_srcimg,
                           c->d_comps, c->d_work_total, (int)rec_bound, d_v, d_d, d_h, d_counters, d_status);
        if (c->timing) { HIP_TRY(hipEventRecord(c->ev[2], stream)); HIP_TRY(hipEventRecord(c->ev[3], stream)); c->ev_valid = 1; c->ev_split = 0; }
        HIP_TRY(hipGetLastError());
        return CELESTE_OK;
    }
    const dim3 grid((unsigned)std::max<size_t>(grid_need, 1));
#define PIXEL_ARGS                                                                                                \
    c->d_images, c->d_patches, c->d_coefs, c->d_bitmaps, c->d_srcimg, c->d_comps, c->d_nbr_off, c->d_nbr_idx,       \
    c->d_val_off, c->d_val, d_targets, c->N, c->NC, CH, chunk_px, G, c->d_acc, c->d_tile_off, c->d_rec, \
    d_active_rank, c->d_items, c->M, c->d_work, c->d_work_total, c->d_nv_base, c->d_nbr_vis, c->d_rec_off, c->d_coefs_f
#define LAUNCH_PIXEL_T(MODE, R) hipLaunchKernelGGL((pixel_kernel<MODE, R>), grid, dim3(64), 0, stream, PIXEL_ARGS)
#define LAUNCH_PIXEL_M(MODE) hipLaunchKernelGGL((pixel_kernel<MODE, double, true>), grid, dim3(64), 0, stream, PIXEL_ARGS)
#define LAUNCH_PIXEL(MODE) do { if (flags & CELESTE_FLAG_FP32) LAUNCH_PIXEL_T(MODE, float); else LAUNCH_PIXEL_T(MODE, double); } while (0)
    if (d_active_rank) {   // several active sources: fp64, fused
        if (flags & CELESTE_FLAG_HESS) LAUNCH_PIXEL_M(2);
        else if (derivs) LAUNCH_PIXEL_M(1);
        else LAUNCH_PIXEL_M(0);
    }
    else if (split) LAUNCH_PIXEL(3);
    else if (flags & CELESTE_FLAG_HESS) LAUNCH_PIXEL(2);
    else if (derivs) LAUNCH_PIXEL(1);
    else LAUNCH_PIXEL_T(0, double);
#undef LAUNCH_PIXEL
#undef LAUNCH_PIXEL_M
#undef PIXEL_ARGS
#undef LAUNCH_PIXEL_T
    if (c->timing) HIP_TRY(hipEventRecord(c->ev[2], stream));
    c->ev_split = 0;
    // every gradient / Hessian entry is owned by one thread whatever the block size: 512 threads halve the latency of
    // a batch that leaves the chip mostly idle anyway; 256 keep 8 workgroups per CU for a 2000-target sweep
    const int lift_nt = n_targets <= 1024 ? 512 : 256;
    // (beyond one round of 8 workgroups per CU the spill-free instantiation: see lift_kernel; CELESTE_LIFT_WAVES=8 / 7 forces one)
    bool lift7 = n_targets > 4096;
    if (const char *e = getenv("CELESTE_LIFT_WAVES")) lift7 = atoi(e) == 7;
#define LAUNCH_LIFT(...) do { if (lift7) hipLaunchKernelGGL(lift_kernel<7>, __VA_ARGS__); else hipLaunchKernelGGL(lift_kernel<8>, __VA_ARGS__); } while (0)
    if (split) {
        // per-patch sums of the records, then the lift reads one record per (target, image): CH = 1
        // its own (part, patch) grid, part index slowest: streaming order matters more to this kernel than the idle
        // workgroups do (run over the pixel kernel's longest-first work list it lost 5 %)
        hipLaunchKernelGGL(record_sum_kernel, dim3((unsigned)((size_t)n_targets * c->M * c->RCH)), dim3(RSUM_NT), 0,
                           stream, c->d_patches, d_targets, c->d_tile_off, reinterpret_cast<const double2 *>(c->d_rec),
                           c->d_items, c->N, c->M, c->RCH, c->sum_tiles, c->d_acc_split, nullptr, c->d_work_total);
        if (c->timing) { HIP_TRY(hipEventRecord(c->ev[4], stream)); c->ev_split = 1; }
        LAUNCH_LIFT(dim3(n_targets), dim3(lift_nt), 0, stream, d_vp, c->d_images, c->d_patches, c->d_geo,
                           c->d_nbr_off, c->d_nbr_idx, d_targets, c->d_acc_split, c->d_prior, c->d_vis_off, c->d_vis_img, c->N, c->M,
                           c->RCH, c->sum_tiles * 64, flags,
                           d_v, d_d, d_h, d_counters, d_status, d_live, d_active_rank ? nullptr : c->d_lg_sum, nullptr, d_h_pos);
    } else
    LAUNCH_LIFT(dim3(n_targets), dim3(lift_nt), 0, stream, d_vp, c->d_images, c->d_patches, c->d_geo,
                       c->d_nbr_off, c->d_nbr_idx, d_targets, c->d_acc, c->d_prior, c->d_vis_off, c->d_vis_img, c->N, c->M, CH,
                       chunk_px, flags,
                       d_v, d_d, d_h, d_counters, d_status, d_live, d_active_rank ? nullptr : c->d_lg_sum, c->d_rec_off, d_h_pos);
#undef LAUNCH_LIFT
    if (c->timing) { HIP_TRY(hipEventRecord(c->ev[3], stream)); c->ev_valid = 1; }
    HIP_TRY(hipGetLastError());
    return CELESTE_OK;
}

// ---- a copy stream whose copies really run beside the kernels -------------------------------------------------------
// HIP maps streams onto a few hardware queues in the order of their first use, and what else a queue carries decides
// whether a device-to-host copy on the copy stream overlaps a kernel on `stream` at all: one context in four or so drew a
// stream whose copies waited for the kernels (2.0 - 2.4 ms per host-pointer sweep of 2000 sources instead of 1.35;
// profiles/r06_stream_lottery.txt), and creating the stream at another priority only moved the draw.  So the first batch
// that relies on the overlap MEASURES it: a 4 MB copy on the candidate stream against a 300 us spin on `stream`; a
// candidate whose copy ends inside the spin is kept, otherwise up to three fresh streams are tried and the quickest wins.
// About a millisecond, once per context (CELESTE_COPY_STREAM_CHECK=0 skips it).
__global__ void spin_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
static int pick_copy_stream(celeste_ctx_t *c) {
    if (c->copy_stream_checked) return CELESTE_OK;
    c->copy_stream_checked = true;
    if (const char *e = getenv("CELESTE_COPY_STREAM_CHECK")) if (atoi(e) == 0) return CELESTE_OK;
    const size_t bytes = 4u << 20;
    void *d_buf = nullptr, *h_buf = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipMalloc(&d_buf, bytes) != hipSuccess || hipHostMalloc(&h_buf, bytes, hipHostMallocDefault) != hipSuccess ||
        hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        (void)hipGetLastError();
        if (d_buf) (void)hipFree(d_buf);
        if (h_buf) (void)hipHostFree(h_buf);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        return CELESTE_OK;       // (no measurement: the stream stays as it is)
    }
    const float spin_ms = 0.3f;
    auto measure = [&](hipStream_t s, float *ms) -> bool {   // copy end relative to the spin's start
        if (hipEventRecord(e0, c->stream) != hipSuccess) return false;
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, c->stream, (long long)(spin_ms * 1e5));   // wall_clock64: 100 MHz
        if (hipStreamWaitEvent(s, e0, 0) != hipSuccess || hipMemcpyAsync(h_buf, d_buf, bytes, hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipEventRecord(e1, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess)
            return false;
        return hipEventElapsedTime(ms, e0, e1) == hipSuccess;
    };
    float best = 0;
    hipStream_t retired[3]; int n_retired = 0;   // (back to the pool only at the end: a retired stream must not be drawn again here)
    bool ok = measure(c->copy_stream, &best) && measure(c->copy_stream, &best);     // (the first use of a stream creates its queue)
    for (int k = 0; ok && best > spin_ms && k < 3; ++k) {
        hipStream_t s = nullptr;
        float ms = 0;
        if (stream_acquire(c->device, &s) != hipSuccess) break;
        if (measure(s, &ms) && measure(s, &ms) && ms < best) { std::swap(s, c->copy_stream); best = ms; }
        retired[n_retired++] = s;
    }
    for (int k = 0; k < n_retired; ++k) stream_retire(c->device, retired[k]);
    (void)hipGetLastError();
    (void)hipFree(d_buf); (void)hipHostFree(h_buf); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return CELESTE_OK;
}

// ---- page-locked host memory ---------------------------------------------------------------------------------
extern "C" void *celeste_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, std::max<size_t>(bytes, 1), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
extern "C" void celeste_host_free(void *ptr) { if (ptr) (void)hipHostFree(ptr); }
extern "C" int celeste_host_register(void *ptr, size_t bytes) try {
    if (!ptr || bytes == 0) return CELESTE_ERR_INVALID_ARG;
    if (hipHostRegister(ptr, bytes, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); return CELESTE_ERR_HIP; }
    return CELESTE_OK;
} ABI_CATCH
extern "C" int celeste_host_unregister(void *ptr) try {
    if (!ptr) return CELESTE_ERR_INVALID_ARG;
    if (hipHostUnregister(ptr) != hipSuccess) { (void)hipGetLastError(); return CELESTE_ERR_HIP; }
    return CELESTE_OK;
} ABI_CATCH
// is [ptr, ptr + bytes) page-locked memory the DMA engines can write directly?
static bool is_pinned(const void *ptr, size_t bytes) {
    if (!ptr || bytes == 0) return false;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, ptr) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (a.type != hipMemoryTypeHost) return false;
    if (hipPointerGetAttributes(&a, (const char *)ptr + bytes - 1) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

template <class T>
static int pinned_grow(T **p, size_t n) {
    if (*p) { staging_free(*p); *p = nullptr; }
    HIP_TRY(staging_alloc((void **)p, std::max<size_t>(n, 1) * sizeof(T)));
    return CELESTE_OK;
}

// A few targets per call (the literal drop-in: one elbo() per call): two copies instead of seven and no second stream --
// the table and the target list go up as ONE block, every output comes down as ONE block on the context's stream.
#define EVAL_SMALL_MAX 32
static int eval_small(celeste_ctx_t *c, const double *vp, int32_t n_targets, const int32_t *targets, uint32_t flags, double *v,
                      double *d, double *h, int64_t *counters, int32_t *status) {
    const size_t n = (size_t)n_targets, HS = (flags & CELESTE_FLAG_PACKED_HESS) ? CELESTE_HP : (size_t)CEL_P * CEL_P;
    const size_t vp_n = (size_t)c->S * CEL_P;
    const size_t in_n = vp_n + EVAL_SMALL_MAX / 2;                                     // doubles: table, then 32 int32
    const size_t per = 1 + CEL_P + (size_t)CEL_P * CEL_P + 2 + 1, out_n = EVAL_SMALL_MAX * per;
    if (!c->d_small_in) HIP_TRY(hipMalloc((void **)&c->d_small_in, in_n * sizeof(double)));
    if (!c->p_small_in) HIP_TRY(staging_alloc((void **)&c->p_small_in, in_n * sizeof(double)));
    if (!c->d_small_out) HIP_TRY(hipMalloc((void **)&c->d_small_out, out_n * sizeof(double)));
    if (!c->p_small_out) HIP_TRY(staging_alloc((void **)&c->p_small_out, out_n * sizeof(double)));
    memcpy(c->p_small_in, vp, vp_n * sizeof(double));
    memcpy(c->p_small_in + vp_n, targets, n * sizeof(int32_t));
    HIP_TRY(hipMemcpyAsync(c->d_small_in, c->p_small_in, (vp_n + (n + 1) / 2) * sizeof(double), hipMemcpyHostToDevice, c->stream));
    // the block of outputs: v[n], d[n x 44], h[n x HS], counters[n x 2], status[n].  The kernels write it straight into the
    // page-locked host block: no copy down (84 -> 81 us per call on a per-source context; CELESTE_SMALL_ZERO_COPY=0: device
    // block + copy)
    static const bool zero_copy = !(getenv("CELESTE_SMALL_ZERO_COPY") && atoi(getenv("CELESTE_SMALL_ZERO_COPY")) == 0);
    double *out_dev = c->d_small_out;
    if (zero_copy) HIP_TRY(hipHostGetDevicePointer((void **)&out_dev, c->p_small_out, 0));
    double *const b_v = out_dev, *const b_d = b_v + n, *const b_h = b_d + n * CEL_P;
    int64_t *const b_c = reinterpret_cast<int64_t *>(b_h + n * HS);
    int32_t *const b_s = reinterpret_cast<int32_t *>(b_c + 2 * n);
    const size_t bytes = (n * (1 + CEL_P + HS + 2)) * sizeof(double) + n * sizeof(int32_t);
    int64_t n_chunks = 0;
    for (int t = 0; t < n_targets; ++t) n_chunks += c->h_src_chunks[targets[t]];
    int st = launch_eval(c, c->d_small_in, n_targets, reinterpret_cast<const int32_t *>(c->d_small_in + vp_n), flags, b_v, b_d, b_h,
                         b_c, b_s, c->stream, true, nullptr, n_chunks);
    if (st != CELESTE_OK) { (void)hipStreamSynchronize(c->stream); return st; }
    if ((!zero_copy && hipMemcpyAsync(c->p_small_out, c->d_small_out, bytes, hipMemcpyDeviceToHost, c->stream) != hipSuccess) ||
        hipStreamSynchronize(c->stream) != hipSuccess) { (void)hipStreamSynchronize(c->stream); return CELESTE_ERR_HIP; }
    const double *const o_v = c->p_small_out, *const o_d = o_v + n, *const o_h = o_d + n * CEL_P;
    const int64_t *const o_c = reinterpret_cast<const int64_t *>(o_h + n * HS);
    const int32_t *const o_s = reinterpret_cast<const int32_t *>(o_c + 2 * n);
    if (v) memcpy(v, o_v, n * sizeof(double));
    if (d && (flags & (CELESTE_FLAG_GRAD | CELESTE_FLAG_HESS))) memcpy(d, o_d, n * CEL_P * sizeof(double));
    if (h && (flags & CELESTE_FLAG_HESS)) memcpy(h, o_h, n * HS * sizeof(double));
    if (counters) memcpy(counters, o_c, n * 2 * sizeof(int64_t));
    int worst = CELESTE_OK;
    for (int t = 0; t < n_targets; ++t) {
        if (status) status[t] = o_s[t];
        if (o_s[t] != CELESTE_OK && worst == CELESTE_OK) worst = o_s[t];
    }
    return worst;
}

// The host-pointer sweep (what the Julia shim calls).  vp and the target list go up through page-locked staging.  A page-locked
// Hessian array (celeste_host_alloc / celeste_host_register) is written by the lift kernel itself through its device address:
// one launch chain, nothing to copy.  Otherwise the batch is cut into up to MAX_PARTS parts that are evaluated back to back on
// the context's stream while each finished part's results are copied down on the copy stream into page-locked staging,
// followed by a host copy of that part (which overlaps the device work of the following parts).  Only the context's own
// streams are waited for.
extern "C" int celeste_elbo_eval_batch(celeste_ctx_t *c, const double *vp, int32_t n_targets, const int32_t *targets,
                                       uint32_t flags, double *v, double *d, double *h, int64_t *counters,
                                       int32_t *status) try {
    if (!c || !vp || !targets || n_targets < 0) return CELESTE_ERR_INVALID_ARG;
    if (n_targets == 0) return CELESTE_OK;
    for (int t = 0; t < n_targets; ++t) if (targets[t] < 0 || targets[t] >= c->S) return CELESTE_ERR_INVALID_ARG;
    HIP_TRY(hipSetDevice(c->device));
    if (n_targets <= EVAL_SMALL_MAX && !c->timing && v) return eval_small(c, vp, n_targets, targets, flags, v, d, h, counters, status);
    const bool want_h = h && (flags & CELESTE_FLAG_HESS);
    const bool want_d = d && (flags & (CELESTE_FLAG_GRAD | CELESTE_FLAG_HESS));
    const size_t HS = (flags & CELESTE_FLAG_PACKED_HESS) ? CELESTE_HP : (size_t)CEL_P * CEL_P;   // doubles per Hessian
    const size_t n = (size_t)n_targets;
    const size_t vp_bytes = (size_t)c->S * CEL_P * sizeof(double);
    if (!c->d_vp) HIP_TRY(hipMalloc((void **)&c->d_vp, vp_bytes));
    if (!c->p_vp) { int st = pinned_grow(&c->p_vp, (size_t)c->S * CEL_P); if (st != CELESTE_OK) return st; }
    if (n > c->stage_cap) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        void **ps[] = {(void **)&c->d_targets, (void **)&c->d_v, (void **)&c->d_d, (void **)&c->d_h, (void **)&c->d_cnt, (void **)&c->d_status};
        for (void **p : ps) if (*p) { HIP_TRY(hipFree(*p)); *p = nullptr; }
        c->stage_cap = 0;
        HIP_TRY(hipMalloc((void **)&c->d_targets, n * sizeof(int32_t)));
        HIP_TRY(hipMalloc((void **)&c->d_v, n * sizeof(double)));
        HIP_TRY(hipMalloc((void **)&c->d_d, n * CEL_P * sizeof(double)));
        HIP_TRY(hipMalloc((void **)&c->d_h, n * CEL_P * CEL_P * sizeof(double)));
        HIP_TRY(hipMalloc((void **)&c->d_cnt, n * 2 * sizeof(int64_t)));
        HIP_TRY(hipMalloc((void **)&c->d_status, n * sizeof(int32_t)));
        c->stage_cap = n;
    }
    // which outputs need staging on the host
    const bool pin_v = is_pinned(v, n * sizeof(double)), pin_d = !want_d || is_pinned(d, n * CEL_P * sizeof(double));
    const bool pin_h = !want_h || is_pinned(h, n * HS * sizeof(double));
    const bool pin_c = !counters || is_pinned(counters, n * 2 * sizeof(int64_t));
    if (n > c->pin_cap) {
        c->pin_cap = 0;
        int st = pinned_grow(&c->p_targets, n);
        if (st == CELESTE_OK) st = pinned_grow(&c->p_status, n);
        if (st == CELESTE_OK) st = pinned_grow(&c->p_v, n);
        if (st == CELESTE_OK) st = pinned_grow(&c->p_d, n * CEL_P);
        if (st == CELESTE_OK) st = pinned_grow(&c->p_cnt, n * 2);
        if (st == CELESTE_OK) st = pinned_grow(&c->p_h, n * CEL_P * CEL_P);
        if (st != CELESTE_OK) return st;
        c->pin_cap = n;
    }
    // inputs: through page-locked staging unless the caller's vp already is page-locked
    const double *vp_src = vp;
    if (!is_pinned(vp, vp_bytes)) { memcpy(c->p_vp, vp, vp_bytes); vp_src = c->p_vp; }
    memcpy(c->p_targets, targets, n * sizeof(int32_t));
    HIP_TRY(hipMemcpyAsync(c->d_vp, vp_src, vp_bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->d_targets, c->p_targets, n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));

    // parts: at least 192 targets each so that a part still fills the chip; one part when nothing large comes back
    int n_parts = 1;
    if (want_h) n_parts = (int)std::min<size_t>(celeste_ctx::MAX_PARTS, std::max<size_t>(1, n / 192));
    if (c->timing) n_parts = 1;   // the kernel timers describe one launch
    // Page-locked Hessians: the lift kernel writes them STRAIGHT into the caller's array (the device address of the mapped host
    // block) -- no copy engine, no second stream, no cross-stream dependency per part, whose latency is what made one context in
    // four slow (pick_copy_stream); the PCIe writes are the lift's own (CELESTE_HOST_ZERO_COPY=0: the copied parts).
    static const bool zero_copy_h = !(getenv("CELESTE_HOST_ZERO_COPY") && atoi(getenv("CELESTE_HOST_ZERO_COPY")) == 0);
    double *h_mapped = nullptr;
    if (want_h && pin_h && zero_copy_h && !c->timing && hipHostGetDevicePointer((void **)&h_mapped, h, 0) == hipSuccess && h_mapped) n_parts = 1;
    else { h_mapped = nullptr; (void)hipGetLastError(); }
    if (n_parts > 1) (void)pick_copy_stream(c);
    double *const o_v = pin_v ? v : c->p_v;
    double *const o_d = want_d ? (pin_d ? d : c->p_d) : nullptr;
    double *const o_h = want_h ? (pin_h ? h : c->p_h) : nullptr;
    int64_t *const o_c = counters ? (pin_c ? counters : c->p_cnt) : nullptr;
    // From here on copies into the CALLER's buffers may be in flight on the copy stream: an error return first waits for
    // both streams, so that no DMA writes into memory the caller believes is his again (page-locked blocks are recycled)
#define EB_TRY(expr)                                                                                             \
    do {                                                                                                         \
        hipError_t e__ = (expr);                                                                                 \
        if (e__ != hipSuccess) {                                                                                 \
            fprintf(stderr, "celeste_mi355x: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            (void)hipStreamSynchronize(c->stream); (void)hipStreamSynchronize(c->copy_stream);                   \
            return CELESTE_ERR_HIP;                                                                              \
        }                                                                                                        \
    } while (0)
    int part_lo[celeste_ctx::MAX_PARTS + 1];
    for (int k = 0; k <= n_parts; ++k) part_lo[k] = (int)((int64_t)n_targets * k / n_parts);
    { int st_e = ensure_part_events(c, n_parts); if (st_e != CELESTE_OK) return st_e; }
    for (int k = 0; k < n_parts; ++k) {
        const int lo = part_lo[k], cnt = part_lo[k + 1] - lo;
        int64_t n_chunks = 0;   // the targets are known here: exact size of the pixel kernel's work list
        for (int t = lo; t < lo + cnt; ++t) n_chunks += c->h_src_chunks[targets[t]];
        int st = launch_eval(c, c->d_vp, cnt, c->d_targets + lo, flags, c->d_v + lo, c->d_d + (size_t)lo * CEL_P,
                             h_mapped ? h_mapped + (size_t)lo * HS : c->d_h + (size_t)lo * HS, c->d_cnt + 2 * (size_t)lo,
                             c->d_status + lo, c->stream, true, nullptr, n_chunks, k > 0, nullptr, n_parts > 1);
        if (st != CELESTE_OK) { (void)hipStreamSynchronize(c->stream); (void)hipStreamSynchronize(c->copy_stream); return st; }
        EB_TRY(hipEventRecord(c->part_done[k], c->stream));
        EB_TRY(hipStreamWaitEvent(c->copy_stream, c->part_done[k], 0));
        if (o_h && !h_mapped) EB_TRY(hipMemcpyAsync(o_h + (size_t)lo * HS, c->d_h + (size_t)lo * HS, (size_t)cnt * HS * sizeof(double),
                                                     hipMemcpyDeviceToHost, c->copy_stream));
        if (o_d) EB_TRY(hipMemcpyAsync(o_d + (size_t)lo * CEL_P, c->d_d + (size_t)lo * CEL_P,
                                        (size_t)cnt * CEL_P * sizeof(double), hipMemcpyDeviceToHost, c->copy_stream));
        EB_TRY(hipMemcpyAsync(o_v + lo, c->d_v + lo, (size_t)cnt * sizeof(double), hipMemcpyDeviceToHost, c->copy_stream));
        if (o_c) EB_TRY(hipMemcpyAsync(o_c + 2 * (size_t)lo, c->d_cnt + 2 * (size_t)lo, (size_t)cnt * 2 * sizeof(int64_t),
                                        hipMemcpyDeviceToHost, c->copy_stream));
        EB_TRY(hipMemcpyAsync(c->p_status + lo, c->d_status + lo, (size_t)cnt * sizeof(int32_t), hipMemcpyDeviceToHost,
                               c->copy_stream));
        EB_TRY(hipEventRecord(c->part_copied[k], c->copy_stream));
    }
    for (int k = 0; k < n_parts; ++k) {   // staged outputs: host copy of part k while later parts are still in flight
        const int lo = part_lo[k], cnt = part_lo[k + 1] - lo;
        EB_TRY(hipEventSynchronize(c->part_copied[k]));
        if (v && !pin_v) memcpy(v + lo, c->p_v + lo, (size_t)cnt * sizeof(double));
        if (want_d && !pin_d) memcpy(d + (size_t)lo * CEL_P, c->p_d + (size_t)lo * CEL_P, (size_t)cnt * CEL_P * sizeof(double));
        if (want_h && !pin_h) memcpy(h + (size_t)lo * HS, c->p_h + (size_t)lo * HS, (size_t)cnt * HS * sizeof(double));
        if (counters && !pin_c) memcpy(counters + 2 * (size_t)lo, c->p_cnt + 2 * (size_t)lo, (size_t)cnt * 2 * sizeof(int64_t));
    }
    EB_TRY(hipStreamSynchronize(c->stream));
#undef EB_TRY
    int worst = CELESTE_OK;
    for (int t = 0; t < n_targets; ++t) {
        const int32_t s1 = c->p_status[t];
        if (status) status[t] = s1;
        if (s1 != CELESTE_OK && worst == CELESTE_OK) worst = s1;
    }
    return worst;
} ABI_CATCH

extern "C" int celeste_elbo_eval(celeste_ctx_t *c, const double *vp, int32_t target, uint32_t flags, double *v,
                                 double *d, double *h, int64_t *n_active_px, int64_t *n_inactive_px) try {
    int64_t cnt[2] = {0, 0};
    int32_t st1 = 0;
    double vv = 0;
    int st = celeste_elbo_eval_batch(c, vp, 1, &target, flags, &vv, d, h, cnt, &st1);
    if (v) *v = vv;
    if (n_active_px) *n_active_px = cnt[0];
    if (n_inactive_px) *n_inactive_px = cnt[1];
    return st;
} ABI_CATCH

// ---- elbo() with several active sources (ElboArgs.active_sources, elbo_args.jl:165-211) -------------------
extern "C" int celeste_elbo_eval_multi(celeste_ctx_t *c, const double *vp, int32_t n_active, const int32_t *active,
                                       uint32_t flags, double *v, double *d, double *h, int64_t *n_active_px,
                                       int64_t *n_inactive_px) try {
    if (!c || !vp || !active || n_active < 1 || (flags & (CELESTE_FLAG_SPLIT | CELESTE_FLAG_PACKED_HESS))) return CELESTE_ERR_INVALID_ARG;
    const int Sa = n_active;
    for (int a = 0; a < Sa; ++a) {
        if (active[a] < 0 || active[a] >= c->S) return CELESTE_ERR_INVALID_ARG;
        for (int b = 0; b < a; ++b) if (active[b] == active[a]) return CELESTE_ERR_INVALID_ARG;
    }
    const bool want_hess = (flags & CELESTE_FLAG_HESS) != 0;
    const bool want_grad = want_hess || (flags & CELESTE_FLAG_GRAD) != 0;
    if ((want_grad && !d) || (want_hess && !h)) return CELESTE_ERR_INVALID_ARG;
    HIP_TRY(hipSetDevice(c->device));
    const size_t PT = (size_t)CEL_P * Sa;
    std::vector<double> hv(Sa), hd((size_t)Sa * CEL_P), hh(want_hess ? (size_t)Sa * CEL_P * CEL_P : 0);
    std::vector<int64_t> hcnt((size_t)Sa * 2);
    std::vector<int32_t> hst(Sa), rank((size_t)c->S, -1);
    for (int a = 0; a < Sa; ++a) rank[active[a]] = a;
    // pairs of active sources that light common pixels; the first of a pair must list the second as a neighbour
    std::vector<int32_t> pa, pb, pia, pib;
    if (want_hess)
        for (int ia = 0; ia < Sa; ++ia) for (int ib = ia + 1; ib < Sa; ++ib) {
            auto lists = [&](int s, int t) {
                for (int64_t q = c->h_nbr_off[s]; q < c->h_nbr_off[s + 1]; ++q) if (c->h_nbr_idx[q] == t) return true;
                return false;
            };
            if (lists(active[ia], active[ib])) { pa.push_back(active[ia]); pb.push_back(active[ib]); pia.push_back(ia); pib.push_back(ib); }
            else if (lists(active[ib], active[ia])) { pa.push_back(active[ib]); pb.push_back(active[ia]); pia.push_back(ib); pib.push_back(ia); }
        }
    const size_t np = pa.size();
    double *d_vp = nullptr, *d_v = nullptr, *d_d = nullptr, *d_h = nullptr, *d_rec = nullptr, *d_x = nullptr;
    int32_t *d_t = nullptr, *d_st = nullptr, *d_rank = nullptr, *d_pa = nullptr, *d_pb = nullptr;
    int64_t *d_cnt = nullptr;
    int rc = CELESTE_OK;
    std::vector<double> hx(np * LIFT_NP * LIFT_NP);
#define MU_TRY(expr) do { if ((expr) != hipSuccess) { rc = CELESTE_ERR_HIP; goto done; } } while (0)
    MU_TRY(scratch_get(c, 0, (size_t)c->S * CEL_P * sizeof(double), &d_vp));
    MU_TRY(scratch_get(c, 1, Sa * sizeof(double), &d_v));
    MU_TRY(scratch_get(c, 2, (size_t)Sa * CEL_P * sizeof(double), &d_d));
    MU_TRY(scratch_get(c, 3, (size_t)Sa * CEL_P * CEL_P * sizeof(double), &d_h));
    MU_TRY(scratch_get(c, 4, Sa * sizeof(int32_t), &d_t));
    MU_TRY(scratch_get(c, 5, Sa * sizeof(int32_t), &d_st));
    MU_TRY(scratch_get(c, 6, (size_t)Sa * 2 * sizeof(int64_t), &d_cnt));
    MU_TRY(scratch_get(c, 7, (size_t)c->S * sizeof(int32_t), &d_rank));
    MU_TRY(hipMemcpyAsync(d_vp, vp, (size_t)c->S * CEL_P * sizeof(double), hipMemcpyHostToDevice, c->stream));
    MU_TRY(hipMemcpyAsync(d_t, active, Sa * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    MU_TRY(hipMemcpyAsync(d_rank, rank.data(), (size_t)c->S * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    rc = launch_eval(c, d_vp, Sa, d_t, flags, d_v, d_d, d_h, d_cnt, d_st, c->stream, true, d_rank);
    if (rc != CELESTE_OK) goto done;
    if (np > 0) {
        MU_TRY(scratch_get(c, 8, np * sizeof(int32_t), &d_pa));
        MU_TRY(scratch_get(c, 9, np * sizeof(int32_t), &d_pb));
        MU_TRY(scratch_get(c, 10, np * c->N * ZV * ZV * sizeof(double), &d_rec));
        MU_TRY(scratch_get(c, 11, np * LIFT_NP * LIFT_NP * sizeof(double), &d_x));
        MU_TRY(hipMemcpyAsync(d_pa, pa.data(), np * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        MU_TRY(hipMemcpyAsync(d_pb, pb.data(), np * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(cross_kernel, dim3((unsigned)(np * c->N)), dim3(64), 0, c->stream, c->d_images, c->d_patches,
                           c->d_coefs, c->d_bitmaps, c->d_srcimg, c->d_comps, c->d_nbr_off, c->d_nbr_idx, c->d_val_off,
                           c->d_val, d_pa, d_pb, c->N, c->NC, d_rec, c->d_vis_off, c->d_vis_img, (int)c->dense);
        hipLaunchKernelGGL(cross_lift_kernel, dim3((unsigned)np), dim3(256), 0, c->stream, d_vp, c->d_images, c->d_patches,
                           c->d_geo, d_pa, d_pb, d_rec, c->N, d_x, c->d_vis_off, c->d_vis_img, (int)c->dense);
        MU_TRY(hipMemcpyAsync(hx.data(), d_x, hx.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    }
    MU_TRY(hipMemcpyAsync(hv.data(), d_v, Sa * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    MU_TRY(hipMemcpyAsync(hst.data(), d_st, Sa * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    MU_TRY(hipMemcpyAsync(hcnt.data(), d_cnt, hcnt.size() * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    if (want_grad) MU_TRY(hipMemcpyAsync(hd.data(), d_d, hd.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (want_hess) MU_TRY(hipMemcpyAsync(hh.data(), d_h, hh.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    MU_TRY(hipStreamSynchronize(c->stream));
    {
        double vs = 0; int64_t na = 0, ni = 0;
        for (int a = 0; a < Sa; ++a) {
            vs += hv[a]; na += hcnt[2 * a]; ni += hcnt[2 * a + 1];
            if (hst[a] != CELESTE_OK && rc == CELESTE_OK) rc = hst[a];
        }
        if (v) *v = vs;
        if (n_active_px) *n_active_px = na;
        if (n_inactive_px) *n_inactive_px = ni;
        if (want_grad) memcpy(d, hd.data(), hd.size() * sizeof(double));   // P x Sa, column per active source
        if (want_hess) {
            memset(h, 0, PT * PT * sizeof(double));
            for (int a = 0; a < Sa; ++a)
                for (int j = 0; j < CEL_P; ++j) for (int i = 0; i < CEL_P; ++i)
                    h[(CEL_P * a + i) + PT * (CEL_P * a + j)] = hh[(size_t)a * CEL_P * CEL_P + i + CEL_P * j];
            for (size_t k = 0; k < np; ++k)
                for (int p2 = 0; p2 < LIFT_NP; ++p2) for (int p1 = 0; p1 < LIFT_NP; ++p1) {
                    const double x = hx[k * LIFT_NP * LIFT_NP + p1 + LIFT_NP * p2];   // d2 / d theta_a[p1] d theta_b[p2]
                    if (!std::isfinite(x) && rc == CELESTE_OK) rc = CELESTE_ERR_NONFINITE_RESULT;
                    h[(CEL_P * pia[k] + p1) + PT * (CEL_P * pib[k] + p2)] = x;
                    h[(CEL_P * pib[k] + p2) + PT * (CEL_P * pia[k] + p1)] = x;
                }
        }
    }
done:
#undef MU_TRY
    (void)hipStreamSynchronize(c->stream);   // nothing of this call is in flight when it returns
    return rc;
} ABI_CATCH

extern "C" int celeste_ctx_enable_timing(celeste_ctx_t *c, int enable) try {
    if (!c) return CELESTE_ERR_INVALID_ARG;
    if (enable && !c->ev[4]) {      // (created on first use; the caller's current device is left as it was)
        int prev = -1;
        (void)hipGetDevice(&prev);
        HIP_TRY(hipSetDevice(c->device));
        for (int i = 0; i < 5; ++i) if (!c->ev[i]) HIP_TRY(hipEventCreate(&c->ev[i]));
        if (prev >= 0 && prev != c->device) HIP_TRY(hipSetDevice(prev));
    }
    c->timing = enable ? 1 : 0;
    c->ev_valid = 0;
    return CELESTE_OK;
} ABI_CATCH

extern "C" int celeste_ctx_last_kernel_ms(celeste_ctx_t *c, float ms[3]) try {
    if (!c || !ms || !c->ev_valid) return CELESTE_ERR_INVALID_ARG;
    HIP_TRY(hipEventSynchronize(c->ev[3]));
    for (int i = 0; i < 3; ++i) HIP_TRY(hipEventElapsedTime(&ms[i], c->ev[i], c->ev[i + 1]));
    if (c->ev_split) HIP_TRY(hipEventElapsedTime(&ms[2], c->ev[4], c->ev[3]));  // lift alone
    return CELESTE_OK;
} ABI_CATCH

extern "C" int celeste_ctx_last_record_sum_ms(celeste_ctx_t *c, float *ms) try {
    if (!c || !ms || !c->ev_valid || !c->ev_split) return CELESTE_ERR_INVALID_ARG;
    HIP_TRY(hipEventSynchronize(c->ev[3]));
    HIP_TRY(hipEventElapsedTime(ms, c->ev[2], c->ev[4]));
    return CELESTE_OK;
} ABI_CATCH

extern "C" int celeste_ctx_work_stats(celeste_ctx_t *c, int32_t n_targets, const int32_t *targets,
                                      celeste_work_stats_t *out) try {
    if (!c || !out || (n_targets > 0 && !targets)) return CELESTE_ERR_INVALID_ARG;
    memset(out, 0, sizeof *out);
    out->n_targets = n_targets;
    for (int q = 0; q < n_targets; ++q) {
        const int t = targets[q];
        if (t < 0 || t >= c->S) return CELESTE_ERR_INVALID_ARG;
        int64_t A = 0, R = 0, PC = 0;   // PC: non-empty (source, image) patches whose constants are read
        for (int v = c->h_vis_off[t]; v < c->h_vis_off[t + 1]; ++v) {
            const DevPatch &p = c->h_patches[v];
            if (p.H2 * p.W2 > 0) {
                A += (int64_t)p.H2 * p.W2;  // upper bound of visited pixels (NaN / masked pixels are skipped)
                R += p.H2;
                PC += 1;
                out->record_bytes += (int64_t)ACC_N * 8 * ((int64_t)p.H2 * p.W2 + 1);
                out->record_tiles += ((int64_t)p.H2 * p.W2 + 63) / 64;
            }
        }
        const int64_t Kn = c->h_nbr_off[t + 1] - c->h_nbr_off[t];
        for (int64_t q2 = c->h_nbr_off[t]; q2 < c->h_nbr_off[t + 1]; ++q2) {
            const int s2 = c->h_nbr_idx[q2];
            for (int v = c->h_vis_off[s2]; v < c->h_vis_off[s2 + 1]; ++v) PC += c->h_patches[v].H2 * c->h_patches[v].W2 > 0;
        }
        out->active_pixel_visits += A;
        out->patch_rows += R;
        out->neighbor_links += Kn;
        // 9 A + 4 R + 352 (1 + K) + 200 per (source, image) patch of the target and its neighbours
        // (= 200 N (1 + K) when every source is in every image) + 8288
        out->algorithmic_bytes += 9 * A + 4 * R + 352 * (1 + Kn) + 200 * PC + 8288;
    }
    return CELESTE_OK;
} ABI_CATCH

extern "C" int celeste_ctx_spline_coefficients(celeste_ctx_t *c, int32_t stamp, double *coef53) try {
    if (!c || !coef53 || stamp < 0 || stamp >= c->n_stamps) return CELESTE_ERR_INVALID_ARG;
    int st = select_device(c->device);
    if (st != CELESTE_OK) return st;
    HIP_TRY(hipMemcpy(coef53, c->d_coefs + (size_t)stamp * CEL_COEF * CEL_COEF, sizeof(double) * CEL_COEF * CEL_COEF, hipMemcpyDeviceToHost));
    return CELESTE_OK;
} ABI_CATCH

extern "C" int celeste_psf_raster(int device, const double *psf, int32_t K, const double *rows, int32_t n_rows,
                                  const double *cols, int32_t n_cols, double *out) try {
    if (!psf || K <= 0 || !rows || !cols || n_rows <= 0 || n_cols <= 0 || !out) return CELESTE_ERR_INVALID_ARG;
    int st = select_device(device);
    if (st != CELESTE_OK) return st;
    double *d_psf = nullptr, *d_rows = nullptr, *d_cols = nullptr, *d_out = nullptr;
    st = dev_upload(&d_psf, psf, (size_t)K * 6);
    if (st == CELESTE_OK) st = dev_upload(&d_rows, rows, (size_t)n_rows);
    if (st == CELESTE_OK) st = dev_upload(&d_cols, cols, (size_t)n_cols);
    if (st == CELESTE_OK) st = dev_upload<double>(&d_out, nullptr, (size_t)n_rows * n_cols);
    if (st == CELESTE_OK) {
        const int n = n_rows * n_cols;
        hipLaunchKernelGGL(psf_raster_kernel, dim3((n + 255) / 256), dim3(256), 0, nullptr, d_psf, K, d_rows, n_rows,
                           d_cols, n_cols, d_out);
        if (hipMemcpy(out, d_out, (size_t)n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) st = CELESTE_ERR_HIP;
    }
    (void)hipFree(d_psf); (void)hipFree(d_rows); (void)hipFree(d_cols); (void)hipFree(d_out);
    return st;
} ABI_CATCH


// ---- maximize! for a batch of targets (ElboMaximize.jl:228-242; neighbours frozen at the input vp) ----------
// Two drivers of the same device functions:
//  * fused (optim_fused_kernel, fused_kernels.h): ONE persistent launch in which every target iterates at its own pace --
//    the default for batches of up to FUSED_AUTO_MAX targets (Cyclades layers, a rank's shard at N >= 4), where the chained
//    driver's four dependent launches per Newton iteration leave the chip idle;
//  * chained: work list -> pixel_kernel -> lift_kernel -> optim_step_kernel per Newton iteration, all targets in
//    lock-step, for large batches that fill the chip kernel by kernel.  Its Newton loop is device resident: the host
//    enqueues iteration `it` while the device is still two iterations behind, sizing the grids by the number of
//    unconverged targets two iterations ago (an upper bound: the count never grows); the true count lives in device
//    memory and the kernels that depend on it read it there.  The counts come back through page-locked slots with an
//    event each, so the host never waits for the iteration it just enqueued.
// Results are bit-identical (tests/test_gpu_fused.py).  CELESTE_OPT_FUSED=0 / 1 forces one or the other.
#define FUSED_SMALL_TARGETS 160    // batches up to this size run the fused launch with one workgroup per CU (measured: 96 ... 256 alike, 400 loses)
#define FUSED_AUTO_MAX 880          // (measured, config 3, end of round 4: 500 targets 9.1 vs 11.2 ms chained, 750: 12.5 vs 13.1, 1000: 16.2 vs 15.6)
#define JOINT_DATAFLOW_WIDEST 4096  // widest layer of a schedule that still runs as one dataflow launch (measured: 3821 -> 0.158 s against 0.185 layer by layer; 14 782 -> 0.64 against 0.52)

static int optim_config(const celeste_optim_config_t *cfg_in, OptParams *op, uint32_t *flags) {
    celeste_optim_config_t cfg = {1e-4, 1.0, 50, 1, 1e-7, 1e-6, 1e-8, 1.0, 1e9, 0, 0};
    if (cfg_in) cfg = *cfg_in;
    if (!(cfg.loc_width > 0) || !(cfg.loc_scale > 0) || cfg.max_iters < 0 || cfg.tr_secular_iters < 0)
        return CELESTE_ERR_INVALID_ARG;
    op->loc_width = cfg.loc_width; op->loc_scale = cfg.loc_scale; op->xtol_abs = cfg.xtol_abs; op->ftol_rel = cfg.ftol_rel;
    op->gtol = cfg.gtol; op->initial_delta = cfg.initial_delta; op->delta_hat = cfg.delta_hat; op->max_iters = cfg.max_iters;
    op->secular_iters = cfg.tr_secular_iters > 0 ? cfg.tr_secular_iters : 20;
    // CELESTE_TR_SOLVER=eig: full eigen-decomposition for every sub-problem (cross-check of the default
    // tridiagonal-space solve)
    const char *env_solver = getenv("CELESTE_TR_SOLVER");
    op->solver = (env_solver && strcmp(env_solver, "eig") == 0) ? 1 : 0;
    op->pad = 0;
    *flags = CELESTE_FLAG_GRAD | CELESTE_FLAG_HESS | (cfg.include_kl ? CELESTE_FLAG_KL : 0);
    return CELESTE_OK;
}

// per-target buffers of the optimiser at capacity >= n
static int optim_buffers(celeste_ctx_t *c, size_t n, hipStream_t stream) {
    auto &ob = c->opt;
    const size_t vp_bytes = (size_t)c->S * CEL_P * sizeof(double);
    if (n > ob.cap) {   // (re)allocate every per-target buffer at the new capacity
        void **grow[] = {(void **)&ob.d_v, (void **)&ob.d_d, (void **)&ob.d_h, (void **)&ob.d_H, (void **)&ob.d_pos,
                         (void **)&ob.d_targets, (void **)&ob.d_act[0], (void **)&ob.d_act[1], (void **)&ob.d_evt[0],
                         (void **)&ob.d_evt[1], (void **)&ob.d_st, &ob.d_state, (void **)&ob.d_T, (void **)&ob.d_S};
        const size_t bytes[] = {sizeof(double), CEL_P * sizeof(double), CEL_P * CEL_P * sizeof(double), NF * NF * sizeof(double),
                                2 * sizeof(double), sizeof(int32_t), sizeof(int32_t), sizeof(int32_t), sizeof(int32_t),
                                sizeof(int32_t), sizeof(int32_t), sizeof(OptState), TRI_STATE * sizeof(double), 2 * SPEC_STATE * sizeof(double)};
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        ob.cap = 0;
        for (int k = 0; k < 14; ++k) {
            if (*grow[k]) { (void)hipFree(*grow[k]); *grow[k] = nullptr; }
            HIP_TRY(hipMalloc(grow[k], n * bytes[k]));
        }
        if (ob.h_state) { staging_free(ob.h_state); ob.h_state = nullptr; }
        HIP_TRY(staging_alloc(&ob.h_state, n * sizeof(OptState)));
        ob.cap = n;
    }
    if (!ob.d_vp) HIP_TRY(hipMalloc((void **)&ob.d_vp, vp_bytes));
    if (!ob.d_count) HIP_TRY(hipMalloc((void **)&ob.d_count, 3 * sizeof(int32_t)));   // two counters + blocks_done
    if (!ob.h_vp) HIP_TRY(staging_alloc((void **)&ob.h_vp, 2 * vp_bytes));
    if (!ob.h_count) {
        HIP_TRY(staging_alloc((void **)&ob.h_count, celeste_ctx::OptBuffers::RING * sizeof(int32_t)));
        for (int k = 0; k < celeste_ctx::OptBuffers::RING; ++k) HIP_TRY(hipEventCreateWithFlags(&ob.ev[k], hipEventDisableTiming));
    }
    return CELESTE_OK;
}

// the neighbours of the batch's targets as they act during the optimisation: rendered once, from the table at d_table
// (ParallelRun.process_source, ParallelRun.jl:468-498), together with the batch's bookkeeping
static int optim_render(celeste_ctx_t *c, const double *d_table, int32_t n_targets, const int32_t *d_targets,
                        int64_t n_chunks, hipStream_t stream) {
    return launch_eval(c, d_table, n_targets, d_targets, 0, c->opt.d_v, nullptr, nullptr, nullptr, c->opt.d_st, stream, true,
                       nullptr, n_chunks, false, nullptr, false, /*render_only=*/true);
}

// how the batch is optimised: 1 = fused launch, 0 = chained
static bool optim_use_fused(celeste_ctx_t *c, int32_t n_targets, int64_t n_chunks, const OptParams &op, bool any_size = false) {
    int mode = -1;
    if (const char *e = getenv("CELESTE_OPT_FUSED")) mode = atoi(e);
    if (mode == 0) return false;
    const int64_t rec = n_chunks >= 0 ? n_chunks : std::min<int64_t>((int64_t)n_targets * c->max_src_chunks, (int64_t)n_targets * c->M * c->CH);
    const int64_t q = (rec + n_targets) * ((int64_t)op.max_iters + 2) + 4096;
    if (q > (int64_t)1 << 28) return false;   // a queue of more than 1 GiB of tickets: lock-step is the better tool
    if (mode == 1 || any_size) return true;
    return n_targets <= FUSED_AUTO_MAX;
}

// The fused launch.  The targets have been initialised (optim_init_kernel) and rendered (optim_render) on `stream`;
// asynchronous.  c->fused.h_ctl[FQC_ABORT] != 0 after the stream has drained: the launch gave up (see fused_kernels.h).
// joint mode (celeste_joint_infer): the entries' dependency counters and successor lists, per-entry scratch
struct JointLaunch {
    int32_t *d_dep; const int32_t *d_succ_off, *d_succ; int32_t *d_render_arr; double *d_saved; const double *d_pos;
    int gshift; size_t render_groups;   // bits of a render-group index; groups of the whole schedule
};
static int optim_run_fused(celeste_ctx_t *c, double *d_vp, int32_t n_targets, const int32_t *d_targets, int64_t n_chunks,
                           const OptParams &op, uint32_t flags, hipStream_t stream, const JointLaunch *J = nullptr) {
    auto &fb = c->fused;
    auto &ob = c->opt;
    const size_t n = (size_t)n_targets;
    const size_t rec = (size_t)(n_chunks >= 0 ? n_chunks
                                              : std::min<int64_t>((int64_t)n_targets * c->max_src_chunks, (int64_t)n_targets * c->M * c->CH));
    if (!fb.max_resident) {
        int per_cu = 0, cus = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, optim_fused_kernel<true>, FUSED_NT, 0));
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
        fb.max_resident = std::max(1, std::min(per_cu, 2) * cus);
        fb.cus = cus;
        if (const char *e = getenv("CELESTE_FUSED_GRID")) if (atoi(e) > 0) fb.max_resident = atoi(e);
    }
    // A small batch (one layer of a Cyclades schedule: 80 targets, ~900 records) gets ONE workgroup per CU: its wavefronts then
    // have their SIMDs to themselves in every phase (a chunk item takes 6 us instead of 11 next to a second workgroup's), and the
    // launch is latency-bound -- 86.7 -> 81.5 us per Newton iteration of the layer.  Larger batches need the second workgroup's
    // throughput (the joint schedule of config 3: 0.060 s with two per CU, 0.079 s with one; layers of up to 400: 0.183 s against
    // 0.171 layer by layer).  CELESTE_FUSED_SMALL overrides the threshold, CELESTE_FUSED_GRID the grid.
    static const int small_targets = getenv("CELESTE_FUSED_SMALL") ? atoi(getenv("CELESTE_FUSED_SMALL")) : FUSED_SMALL_TARGETS;
    size_t resident = (size_t)fb.max_resident;
    if (!J && n_targets <= small_targets && fb.cus > 0 && !getenv("CELESTE_FUSED_GRID")) resident = std::min<size_t>(resident, (size_t)fb.cus);
    const int G = (int)std::max<size_t>(1, std::min<size_t>(resident, rec + n));
    const size_t q_cap = (rec + n) * ((size_t)op.max_iters + 2) + (J ? n + J->render_groups : 0) + (size_t)G + 64;
    if (q_cap > 0x7fffffffull) return CELESTE_ERR_INVALID_ARG;
    { int stb = fused_buffers(c, n, rec, stream); if (stb != CELESTE_OK) return stb; }
    if (q_cap > fb.cap_q) {
        HIP_TRY(hipStreamSynchronize(stream));
        if (fb.d_q_items) { (void)hipFree(fb.d_q_items); fb.d_q_items = nullptr; }
        fb.cap_q = 0;
        HIP_TRY(hipMalloc((void **)&fb.d_q_items, q_cap * sizeof(int32_t)));
        fb.cap_q = q_cap;
    }
    if (!fb.d_q_ctl) HIP_TRY(hipMalloc((void **)&fb.d_q_ctl, FQC_WORDS * sizeof(int32_t)));
    if (!fb.h_ctl) HIP_TRY(staging_alloc((void **)&fb.h_ctl, FQC_WORDS * sizeof(int32_t)));
    // every polled word is reset before every launch
    HIP_TRY(hipMemsetAsync(fb.d_q_items, 0xFF, q_cap * sizeof(int32_t), stream));
    HIP_TRY(hipMemsetAsync(fb.d_q_ctl, 0, FQC_WORDS * sizeof(int32_t), stream));
    HIP_TRY(hipMemsetAsync(fb.d_arrivals, 0, n * sizeof(int32_t), stream));
    fb.arrivals_dirty = true;      // (clean again only if the launch runs to its end: eval_fused re-checks)
    hipLaunchKernelGGL(fused_setup_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, d_targets, n_targets,
                       c->d_patches, c->d_vis_off, c->dense ? nullptr : c->d_items, c->N, c->M, c->chunk_px, c->d_rec_off,
                       fb.d_chunk_desc, fb.d_tgt_rec, fb.d_q_items, fb.d_q_ctl, J ? J->d_dep : nullptr, (int)rec);
    FusedArgs A;
    fused_args_tables(c, A);
    A.targets = d_targets; A.n_targets = n_targets; A.vp = d_vp;
    A.chunk_desc = fb.d_chunk_desc; A.tgt_rec = fb.d_tgt_rec;
    A.st = (OptState *)ob.d_state; A.Hstate = ob.d_H; A.Tstate = ob.d_T; A.op = op; A.flags = flags;
    A.Spec = ob.d_S;
    if (const char *e = getenv("CELESTE_FUSED_SPEC")) if (atoi(e) == 0) A.Spec = nullptr;
    if (A.Spec) HIP_TRY(hipMemsetAsync(ob.d_S, 0xFF, n * 2 * SPEC_STATE * sizeof(double), stream));   // no tag matches an iteration
    if (J) {
        A.j_R = (int)rec; A.j_gshift = J->gshift; A.j_dep = J->d_dep; A.j_succ_off = J->d_succ_off; A.j_succ = J->d_succ;
        A.j_vitem_off = c->d_vitem_off; A.j_vitems = c->d_vitems_src; A.j_render_arr = J->d_render_arr; A.j_saved = J->d_saved;
        A.j_pos = J->d_pos; A.j_srcimg = c->d_srcimg; A.j_comps = c->d_comps; A.j_geo = c->d_geo;
    }
    A.q_items = fb.d_q_items; A.q_ctl = fb.d_q_ctl; A.arrivals = fb.d_arrivals; A.q_cap = (int)std::min<size_t>(q_cap, 0x7fffffff);
    double tmo_s = 10.0;
    if (const char *e = getenv("CELESTE_FUSED_TIMEOUT_S")) if (atof(e) > 0) tmo_s = atof(e);
    A.timeout_ticks = (long long)(tmo_s * 1e8);   // wall_clock64: 100 MHz
    if (J) hipLaunchKernelGGL(optim_fused_kernel<true>, dim3((unsigned)G), dim3(FUSED_NT), 0, stream, A);
    else hipLaunchKernelGGL(optim_fused_kernel<false>, dim3((unsigned)G), dim3(FUSED_NT), 0, stream, A);
    HIP_TRY(hipMemcpyAsync(fb.h_ctl, fb.d_q_ctl, FQC_WORDS * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipGetLastError());
    return CELESTE_OK;
}

// The chained driver.  Blocks the host until the batch has converged.
static int optim_run_chained(celeste_ctx_t *c, double *d_vp, int32_t n_targets, const int32_t *d_targets, const OptParams &op,
                             uint32_t flags, hipStream_t stream) {
    auto &ob = c->opt;
    const size_t n = (size_t)n_targets;
    double *const d_v = ob.d_v, *const d_d = ob.d_d, *const d_h = ob.d_h, *const d_H = ob.d_H;
    int32_t *const d_st = ob.d_st;
    int32_t *const d_cnt[2] = {ob.d_count, ob.d_count + 1};
    int32_t *const *d_act = ob.d_act, *const *d_evt = ob.d_evt;
    OptState *const d_state = (OptState *)ob.d_state;
    constexpr int RING = celeste_ctx::OptBuffers::RING;
    HIP_TRY(hipMemsetAsync(ob.d_count, 0, 3 * sizeof(int32_t), stream));   // (an earlier call may have stopped mid-loop)
    // both lists start as the full target list: the step kernel only rewrites the first *next_count entries of the other
    // list, and the launches are sized by a count two iterations old -- entries past the live count must be valid ids
    HIP_TRY(hipMemcpyAsync(d_evt[0], d_targets, n * sizeof(int32_t), hipMemcpyDeviceToDevice, stream));
    HIP_TRY(hipMemcpyAsync(d_evt[1], d_targets, n * sizeof(int32_t), hipMemcpyDeviceToDevice, stream));
    int32_t n_upper = n_targets;
    int cur = 0;
    for (int it = 0; it <= op.max_iters + 1; ++it) {
        if (it >= 2) {   // the count two iterations back: an upper bound of the live targets, 0 = all converged
            HIP_TRY(hipEventSynchronize(ob.ev[(it - 2) % RING]));
            n_upper = ob.h_count[(it - 2) % RING];
            if (n_upper <= 0) break;
        }
        const int32_t *d_live = it == 0 ? nullptr : d_cnt[cur];
        int st1 = launch_eval(c, d_vp, n_upper, d_evt[cur], flags, d_v, d_d, d_h, nullptr, d_st, stream, false, nullptr,
                              -1, false, d_live);
        if (st1 != CELESTE_OK) return st1;
        hipLaunchKernelGGL(optim_step_kernel, dim3(n_upper), dim3(64), 0, stream, d_vp, d_targets, d_act[cur],
                           d_v, d_d, d_h, d_st, op, d_state, d_H, d_act[1 - cur], d_evt[1 - cur], d_cnt[1 - cur],
                           it == 0 ? nullptr : d_cnt[cur], ob.d_count + 2, &ob.h_count[it % RING], ob.d_T);
        HIP_TRY(hipEventRecord(ob.ev[it % RING], stream));
        cur = 1 - cur;
    }
    return CELESTE_OK;
}

// maximize! of the targets against the table at d_vp (rendering done): save the rows, initialise, run, write the
// per-target outputs (device pointers, may be NULL) and give failed targets their rows back.  *fused_out: which driver.
static int optim_run(celeste_ctx_t *c, double *d_vp, const double *d_pos, int32_t n_targets, const int32_t *d_targets,
                     int64_t n_chunks, const OptParams &op, uint32_t flags, int32_t *d_iterations, int32_t *d_f_evals,
                     double *d_elbo, int32_t *d_status, hipStream_t stream, bool *fused_out) {
    auto &ob = c->opt;
    auto &fb = c->fused;
    const size_t n = (size_t)n_targets;
    if (n * CEL_P > fb.cap_saved) {
        HIP_TRY(hipStreamSynchronize(stream));
        if (fb.d_saved) { (void)hipFree(fb.d_saved); fb.d_saved = nullptr; }
        fb.cap_saved = 0;
        HIP_TRY(hipMalloc((void **)&fb.d_saved, n * CEL_P * sizeof(double)));
        fb.cap_saved = n * CEL_P;
    }
    hipLaunchKernelGGL(save_rows_kernel, dim3((unsigned)((n * CEL_P + 255) / 256)), dim3(256), 0, stream, d_vp, d_targets,
                       n_targets, fb.d_saved);
    // (no reduced form yet: NaN everywhere -- the first sub-problem of every target starts its eigenvalue search cold)
    HIP_TRY(hipMemsetAsync(ob.d_T, 0xFF, n * TRI_STATE * sizeof(double), stream));
    hipLaunchKernelGGL(optim_init_kernel, dim3((n_targets + 63) / 64), dim3(64), 0, stream, d_vp, d_targets, n_targets, op,
                       (OptState *)ob.d_state, ob.d_act[0], d_pos);
    const bool fused = optim_use_fused(c, n_targets, n_chunks, op);
    if (fused_out) *fused_out = fused;
    int st = fused ? optim_run_fused(c, d_vp, n_targets, d_targets, n_chunks, op, flags, stream)
                   : optim_run_chained(c, d_vp, n_targets, d_targets, op, flags, stream);
    if (st != CELESTE_OK) return st;
    hipLaunchKernelGGL(optim_finalize_kernel, dim3((n_targets + 63) / 64), dim3(64), 0, stream, (const OptState *)ob.d_state,
                       d_targets, n_targets, d_vp, fb.d_saved, d_iterations, d_f_evals, d_elbo, d_status,
                       fused ? fb.d_q_ctl : nullptr);
    HIP_TRY(hipGetLastError());
    return CELESTE_OK;
}

// did the last fused launch on this context give up?  (after the stream has been synchronised)
static int fused_abort_status(celeste_ctx_t *c) {
    const int code = c->fused.h_ctl ? c->fused.h_ctl[FQC_ABORT] : 0;
    if (code == 0) return CELESTE_OK;
    fprintf(stderr, "celeste_mi355x: the fused optimiser launch gave up (%s)\n",
            code == 1 ? "a workgroup waited longer than CELESTE_FUSED_TIMEOUT_S for work" : "queue capacity exceeded");
    return CELESTE_ERR_HIP;
}

extern "C" int celeste_maximize_batch_device(celeste_ctx_t *c, double *d_vp, const double *d_vp_neighbors,
                                             const double *d_pos_centers, int32_t n_targets, const int32_t *d_targets,
                                             const celeste_optim_config_t *cfg_in, int32_t *d_iterations, int32_t *d_f_evals,
                                             double *d_elbo, int32_t *d_status, void *stream_) try {
    if (!c || !d_vp || !d_targets || n_targets < 0) return CELESTE_ERR_INVALID_ARG;
    if (n_targets == 0) return CELESTE_OK;
    OptParams op;
    uint32_t flags;
    int st = optim_config(cfg_in, &op, &flags);
    if (st != CELESTE_OK) return st;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t stream = (hipStream_t)stream_;
    st = optim_buffers(c, (size_t)n_targets, stream);
    if (st != CELESTE_OK) return st;
    st = optim_render(c, d_vp_neighbors ? d_vp_neighbors : d_vp, n_targets, d_targets, -1, stream);
    if (st != CELESTE_OK) return st;
    return optim_run(c, d_vp, d_pos_centers, n_targets, d_targets, -1, op, flags, d_iterations, d_f_evals, d_elbo, d_status,
                     stream, nullptr);
} ABI_CATCH

static int check_distinct_targets(celeste_ctx_t *c, int32_t n, const int32_t *targets, std::vector<uint8_t> &seen) {
    seen.assign((size_t)c->S, 0);   // two optimisations of one source would share its row of vp
    for (int t = 0; t < n; ++t) {
        if (targets[t] < 0 || targets[t] >= c->S || seen[targets[t]]) return CELESTE_ERR_INVALID_ARG;
        seen[targets[t]] = 1;
    }
    return CELESTE_OK;
}

extern "C" int celeste_maximize_batch(celeste_ctx_t *c, double *vp, const double *vp_neighbors,
                                      const double *pos_centers, int32_t n_targets, const int32_t *targets,
                                      const celeste_optim_config_t *cfg_in, int32_t *iterations, int32_t *f_evals,
                                      double *elbo, int32_t *status) try {
    if (!c || !vp || !targets || n_targets < 0) return CELESTE_ERR_INVALID_ARG;
    if (n_targets == 0) return CELESTE_OK;
    std::vector<uint8_t> seen;
    if (check_distinct_targets(c, n_targets, targets, seen) != CELESTE_OK) return CELESTE_ERR_INVALID_ARG;
    OptParams op;
    uint32_t flags;
    int rc = optim_config(cfg_in, &op, &flags);
    if (rc != CELESTE_OK) return rc;
    HIP_TRY(hipSetDevice(c->device));
    const size_t n = (size_t)n_targets;
    const size_t vp_bytes = (size_t)c->S * CEL_P * sizeof(double);
    auto &ob = c->opt;
    hipStream_t stream = c->stream;
    rc = optim_buffers(c, n, stream);
    if (rc != CELESTE_OK) return rc;
    int64_t n_chunks = 0;
    for (int t = 0; t < n_targets; ++t) n_chunks += c->h_src_chunks[targets[t]];
#define MX_TRY(expr) do { if ((expr) != hipSuccess) { rc = CELESTE_ERR_HIP; goto cleanup; } } while (0)
    {
    double *const d_vp = ob.d_vp;
    double *const d_pos = pos_centers ? ob.d_pos : nullptr;
    int32_t *const d_targets = ob.d_targets;
    double *const h_vp0 = ob.h_vp, *const h_vp1 = ob.h_vp + (size_t)c->S * CEL_P;
    MX_TRY(hipMemcpyAsync(d_targets, targets, n * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    if (pos_centers) MX_TRY(hipMemcpyAsync(d_pos, pos_centers, n * 2 * sizeof(double), hipMemcpyHostToDevice, stream));
    // neighbours are rendered once, from their frozen parameters, before any target moves
    memcpy(h_vp0, vp_neighbors ? vp_neighbors : vp, vp_bytes);
    MX_TRY(hipMemcpyAsync(d_vp, h_vp0, vp_bytes, hipMemcpyHostToDevice, stream));
    rc = optim_render(c, d_vp, n_targets, d_targets, n_chunks, stream);
    if (rc != CELESTE_OK) goto cleanup;
    if (vp_neighbors) {
        memcpy(h_vp1, vp, vp_bytes);
        MX_TRY(hipMemcpyAsync(d_vp, h_vp1, vp_bytes, hipMemcpyHostToDevice, stream));
    }
    bool fused = false;
    rc = optim_run(c, d_vp, d_pos, n_targets, d_targets, n_chunks, op, flags, nullptr, nullptr, nullptr, nullptr, stream, &fused);
    if (rc != CELESTE_OK) goto cleanup;
    OptState *const hs = (OptState *)ob.h_state;
    MX_TRY(hipMemcpyAsync(hs, ob.d_state, n * sizeof(OptState), hipMemcpyDeviceToHost, stream));
    MX_TRY(hipMemcpyAsync(h_vp0, d_vp, vp_bytes, hipMemcpyDeviceToHost, stream));
    MX_TRY(hipStreamSynchronize(stream));
    if (fused && (rc = fused_abort_status(c)) != CELESTE_OK) goto cleanup;
    for (int t = 0; t < n_targets; ++t) {
        // a target that failed keeps its input row; the others are unaffected (ParallelRun.jl:582-597)
        if (hs[t].status == CELESTE_OK)
            memcpy(vp + (size_t)targets[t] * CEL_P, h_vp0 + (size_t)targets[t] * CEL_P, CEL_P * sizeof(double));
        if (iterations) iterations[t] = hs[t].iter;
        if (f_evals) f_evals[t] = hs[t].evals;
        if (elbo) elbo[t] = -hs[t].f;
        if (status) status[t] = hs[t].status;
        if (hs[t].status != CELESTE_OK && rc == CELESTE_OK) rc = hs[t].status;
    }
    }
cleanup:
#undef MX_TRY
    if (rc == CELESTE_ERR_HIP) (void)hipStreamSynchronize(stream);
    return rc;
} ABI_CATCH


// ---- joint inference as ONE launch: the schedule's entries as a dataflow (fused_kernels.h, joint mode) ----------------
#define JOINT_DATAFLOW_MAX 262144
// *ran = false (and CELESTE_OK): the schedule does not fit the launch's encodings -- the caller runs it layer by layer
// d_table: the parameter table the schedule runs against (n_sources x 44, HBM); d_restore: a device copy of it as it was on
// entry (the launch may give up half way -- queue capacity -- and the layered driver then starts from the entry state)
static int joint_dataflow(celeste_ctx_t *c, double *d_table, const double *d_restore, int64_t total, const int32_t *targets,
                          const int32_t *d_all, const double *d_pos,
                          const OptParams &op, uint32_t flags, int32_t *d_it, int32_t *d_ev, double *d_el, int32_t *d_stt,
                          hipStream_t stream, bool *ran) {
    *ran = false;
    const int E = (int)total;
    // a flattened schedule too large for one batch's index space (launch_eval's n_targets x M x CH bound) is a limit of THIS
    // driver, not of the call: the caller runs it layer by layer.  Every other CELESTE_ERR_INVALID_ARG below is an error.
    if ((size_t)E * c->M * c->CH > 0x7fffffffull) return CELESTE_OK;
    auto &ob = c->opt;
    auto &fb = c->fused;
    // An entry waits for the last earlier entry that wrote a row it reads -- its source's, its neighbours' -- and for
    // the earlier entries that read the row it writes: the order the layer-by-layer schedule enforces with barriers.
    std::vector<int32_t> last((size_t)c->S, -1), dep((size_t)E, 0), deps;
    std::vector<std::vector<int32_t>> readers((size_t)c->S), succ((size_t)E);
    int64_t n_chunks = 0;
    size_t groups = 0, gmax = 1;
    for (int e = 0; e < E; ++e) {
        const int t = targets[e];
        deps.clear();
        if (last[t] >= 0) deps.push_back(last[t]);
        for (int64_t q = c->h_nbr_off[t]; q < c->h_nbr_off[t + 1]; ++q) if (last[c->h_nbr_idx[q]] >= 0) deps.push_back(last[c->h_nbr_idx[q]]);
        deps.insert(deps.end(), readers[t].begin(), readers[t].end());
        std::sort(deps.begin(), deps.end());
        deps.erase(std::unique(deps.begin(), deps.end()), deps.end());
        for (int32_t d : deps) succ[(size_t)d].push_back(e);
        dep[(size_t)e] = (int32_t)deps.size();
        last[t] = e;
        readers[t].clear();
        for (int64_t q = c->h_nbr_off[t]; q < c->h_nbr_off[t + 1]; ++q) readers[c->h_nbr_idx[q]].push_back(e);
        n_chunks += c->h_src_chunks[t];
        const size_t g = ((size_t)(c->h_vitem_off[t + 1] - c->h_vitem_off[t]) + FUSED_WAVES - 1) / FUSED_WAVES;
        groups += g; gmax = std::max(gmax, g);
    }
    std::vector<int32_t> succ_off((size_t)E + 1, 0), succ_flat;
    size_t most = 0;
    for (int e = 0; e < E; ++e) {
        succ_off[(size_t)e + 1] = succ_off[(size_t)e] + (int32_t)succ[(size_t)e].size();
        succ_flat.insert(succ_flat.end(), succ[(size_t)e].begin(), succ[(size_t)e].end());
        most = std::max(most, succ[(size_t)e].size());
    }
    int gshift = 1;
    while (((size_t)1 << gshift) < gmax) ++gshift;
    // item codes are int32: records, then one START per entry, then (entry, render group); the ready list of an ending
    // entry is staged in 2048 LDS words
    if (most > 2048 || (uint64_t)n_chunks + (uint64_t)E + ((uint64_t)E << gshift) >= 0x7fffffffull) return CELESTE_OK;
    if (!optim_use_fused(c, E, n_chunks, op, /*any_size=*/true)) return CELESTE_OK;
    int32_t *d_dep = nullptr, *d_succ_off = nullptr, *d_succ = nullptr, *d_rarr = nullptr;
    int rc = CELESTE_OK;
#define JD_TRY(expr) do { if ((expr) != hipSuccess) { rc = CELESTE_ERR_HIP; goto out; } } while (0)
    // (kept between calls: a schedule per box is the normal use)
    JD_TRY(scratch_get(c, 19, (size_t)E * sizeof(int32_t), &d_dep));
    JD_TRY(scratch_get(c, 20, ((size_t)E + 1) * sizeof(int32_t), &d_succ_off));
    JD_TRY(scratch_get(c, 21, std::max<size_t>(succ_flat.size(), 1) * sizeof(int32_t), &d_succ));
    JD_TRY(scratch_get(c, 22, (size_t)E * sizeof(int32_t), &d_rarr));
    if ((size_t)E * CEL_P > fb.cap_saved) {
        JD_TRY(hipStreamSynchronize(stream));
        if (fb.d_saved) { (void)hipFree(fb.d_saved); fb.d_saved = nullptr; }
        fb.cap_saved = 0;
        JD_TRY(hipMalloc((void **)&fb.d_saved, (size_t)E * CEL_P * sizeof(double)));
        fb.cap_saved = (size_t)E * CEL_P;
    }
    JD_TRY(hipMemcpyAsync(d_dep, dep.data(), (size_t)E * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    JD_TRY(hipMemcpyAsync(d_succ_off, succ_off.data(), ((size_t)E + 1) * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    if (!succ_flat.empty())
        JD_TRY(hipMemcpyAsync(d_succ, succ_flat.data(), succ_flat.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    JD_TRY(hipMemsetAsync(d_rarr, 0, (size_t)E * sizeof(int32_t), stream));
    // the batch's bookkeeping for all entries (visit items, record offsets), every source's tables and shape derivatives
    // from the input table; the entries refresh them as they end
    rc = optim_render(c, d_table, E, d_all, n_chunks, stream);
    if (rc != CELESTE_OK) goto out;
    {
        JD_TRY(hipMemsetAsync(ob.d_T, 0xFF, (size_t)E * TRI_STATE * sizeof(double), stream));
        JointLaunch J = {d_dep, d_succ_off, d_succ, d_rarr, fb.d_saved, d_pos, gshift, groups};
        rc = optim_run_fused(c, d_table, E, d_all, n_chunks, op, flags, stream, &J);
        if (rc != CELESTE_OK) goto out;
        // (failed entries gave their rows back in flight: no saved rows here)
        hipLaunchKernelGGL(optim_finalize_kernel, dim3((unsigned)((E + 63) / 64)), dim3(64), 0, stream, (const OptState *)ob.d_state,
                           d_all, E, d_table, (const double *)nullptr, d_it, d_ev, d_el, d_stt, fb.d_q_ctl);
        JD_TRY(hipGetLastError());
        JD_TRY(hipStreamSynchronize(stream));      // (the host vectors above are in flight until here)
        if (fb.h_ctl && fb.h_ctl[FQC_ABORT] == 2) {
            // the launch ran out of queue capacity (a time-out stays an error: something is wrong with the device): the
            // table is partly optimised -- put the entry state back and let the layered driver run the schedule
            JD_TRY(hipMemcpyAsync(d_table, d_restore, (size_t)c->S * CEL_P * sizeof(double), hipMemcpyDefault, stream));
            fb.h_ctl[FQC_ABORT] = 0;               // (the layered driver's own launches set it again if THEY give up)
            goto out;
        }
        *ran = true;
    }
out:
#undef JD_TRY
    if (rc != CELESTE_OK) (void)hipStreamSynchronize(stream);
    return rc;
}

// ---- joint inference: a schedule of layers against one device-resident parameter table -------------------------
// The schedule against the table at d_table (n_sources x 44, HBM, updated in place).  d_entry: a device snapshot of the
// table as it is on entry, or nullptr -- then ob.h_vp (page-locked) must hold it.  Blocks until the schedule is done; the
// per-entry outputs are host pointers (may be NULL).
static int joint_run(celeste_ctx_t *c, double *d_table, const double *d_entry, int32_t n_layers, const int64_t *layer_offsets,
                     const int32_t *layer_targets, const double *pos_centers, const celeste_optim_config_t *cfg_in,
                     int32_t *iterations, int32_t *f_evals, double *elbo, int32_t *status, bool validate) {
    const int64_t total = layer_offsets[n_layers];
    size_t widest = 0;
    std::vector<uint8_t> seen;
    std::vector<int64_t> layer_chunks((size_t)n_layers, 0);
    for (int l = 0; l < n_layers; ++l) {
        const int64_t lo = layer_offsets[l], hi = layer_offsets[l + 1];
        if (hi < lo || hi - lo > 0x7fffffff) return CELESTE_ERR_INVALID_ARG;
        if (validate && check_distinct_targets(c, (int32_t)(hi - lo), layer_targets + lo, seen) != CELESTE_OK) return CELESTE_ERR_INVALID_ARG;
        // the sources of a layer are optimised simultaneously: none may be another's neighbour
        for (int64_t e = lo; e < hi; ++e) {
            const int t = layer_targets[e];
            if (validate)
                for (int64_t q = c->h_nbr_off[t]; q < c->h_nbr_off[t + 1]; ++q) if (seen[c->h_nbr_idx[q]]) return CELESTE_ERR_INVALID_ARG;
            layer_chunks[l] += c->h_src_chunks[t];
        }
        widest = std::max(widest, (size_t)(hi - lo));
    }
    if (total == 0) return CELESTE_OK;
    OptParams op;
    uint32_t flags;
    int rc = optim_config(cfg_in, &op, &flags);
    if (rc != CELESTE_OK) return rc;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t stream = c->stream;
    // One dataflow launch for the whole schedule when its layers are of the size the fused launch is the better driver
    // for (Cyclades batches); schedules with huge layers (a greedy colouring of a 30 000-source scene: 15 000 sources in
    // the first) run layer by layer, where the lock-step kernels spend half the chip time per evaluation (measured on
    // config 5's scene: colouring 0.60 s layered / 0.99 s dataflow, Cyclades 2.39 s layered / 0.99 s dataflow).
    // CELESTE_JOINT_DATAFLOW=0 / 1 forces one or the other.
    const bool can_dataflow = total <= JOINT_DATAFLOW_MAX && optim_use_fused(c, 1, 1, op);
    bool dataflow = can_dataflow && widest <= JOINT_DATAFLOW_WIDEST;
    if (const char *e = getenv("CELESTE_JOINT_DATAFLOW")) dataflow = can_dataflow && atoi(e) != 0;
    rc = optim_buffers(c, dataflow ? (size_t)total : widest, stream);
    if (rc != CELESTE_OK) return rc;
    auto &ob = c->opt;
    // the whole schedule and the per-entry outputs live on the device for the duration of the call
    int32_t *d_all = nullptr, *d_it = nullptr, *d_ev = nullptr, *d_stt = nullptr;
    double *d_pos = nullptr, *d_el = nullptr;
    std::vector<int32_t> h_it((size_t)total), h_ev((size_t)total), h_st((size_t)total);
    std::vector<double> h_el((size_t)total);
    bool any_fused = false;
    int abort_rc = CELESTE_OK;
#define JI_TRY(expr) do { if ((expr) != hipSuccess) { rc = CELESTE_ERR_HIP; goto done; } } while (0)
    JI_TRY(scratch_get(c, 13, (size_t)total * sizeof(int32_t), &d_all));
    JI_TRY(scratch_get(c, 14, (size_t)total * sizeof(int32_t), &d_it));
    JI_TRY(scratch_get(c, 15, (size_t)total * sizeof(int32_t), &d_ev));
    JI_TRY(scratch_get(c, 16, (size_t)total * sizeof(int32_t), &d_stt));
    JI_TRY(scratch_get(c, 17, (size_t)total * sizeof(double), &d_el));
    if (pos_centers) JI_TRY(scratch_get(c, 18, (size_t)total * 2 * sizeof(double), &d_pos));
    JI_TRY(hipMemcpyAsync(d_all, layer_targets, (size_t)total * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    if (pos_centers) JI_TRY(hipMemcpyAsync(d_pos, pos_centers, (size_t)total * 2 * sizeof(double), hipMemcpyHostToDevice, stream));
    if (dataflow) {
        // one launch for the whole schedule (fused_kernels.h, joint mode)
        rc = joint_dataflow(c, d_table, d_entry ? d_entry : ob.h_vp, total, layer_targets, d_all, d_pos, op, flags, d_it, d_ev,
                            d_el, d_stt, stream, &dataflow);
        if (rc != CELESTE_OK) goto done;
        any_fused = dataflow;
    }
    if (!dataflow)
    for (int l = 0; l < n_layers; ++l) {
        const int64_t lo = layer_offsets[l];
        const int32_t n = (int32_t)(layer_offsets[l + 1] - lo);
        if (n == 0) continue;
        // every layer sees the table as the layers before it left it (ParallelRun.jl:372-397)
        rc = optim_render(c, d_table, n, d_all + lo, layer_chunks[l], stream);
        if (rc != CELESTE_OK) goto done;
        bool fused = false;
        rc = optim_run(c, d_table, d_pos ? d_pos + 2 * lo : nullptr, n, d_all + lo, layer_chunks[l], op, flags, d_it + lo,
                       d_ev + lo, d_el + lo, d_stt + lo, stream, &fused);
        if (rc != CELESTE_OK) goto done;
        any_fused |= fused;
    }
    JI_TRY(hipMemcpyAsync(h_it.data(), d_it, (size_t)total * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    JI_TRY(hipMemcpyAsync(h_ev.data(), d_ev, (size_t)total * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    JI_TRY(hipMemcpyAsync(h_st.data(), d_stt, (size_t)total * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    JI_TRY(hipMemcpyAsync(h_el.data(), d_el, (size_t)total * sizeof(double), hipMemcpyDeviceToHost, stream));
    JI_TRY(hipStreamSynchronize(stream));
    if (any_fused) abort_rc = fused_abort_status(c);   // (of the last fused layer; an earlier one shows in the statuses)
    for (int64_t e = 0; e < total; ++e) {
        if (h_st[e] == CELESTE_ERR_HIP) abort_rc = CELESTE_ERR_HIP;
        if (iterations) iterations[e] = h_it[e];
        if (f_evals) f_evals[e] = h_ev[e];
        if (elbo) elbo[e] = h_el[e];
        if (status) status[e] = h_st[e];
        if (h_st[e] != CELESTE_OK && rc == CELESTE_OK) rc = h_st[e];
    }
    if (abort_rc != CELESTE_OK) rc = abort_rc;
done:
#undef JI_TRY
    (void)hipStreamSynchronize(stream);
    return rc;
}

extern "C" int celeste_joint_infer(celeste_ctx_t *c, double *vp, int32_t n_layers, const int64_t *layer_offsets,
                                   const int32_t *layer_targets, const double *pos_centers,
                                   const celeste_optim_config_t *cfg_in, int32_t *iterations, int32_t *f_evals,
                                   double *elbo, int32_t *status) try {
    if (!c || !vp || n_layers < 0 || (n_layers > 0 && (!layer_offsets || !layer_targets))) return CELESTE_ERR_INVALID_ARG;
    if (n_layers == 0) return CELESTE_OK;
    if (layer_offsets[0] != 0) return CELESTE_ERR_INVALID_ARG;
    if (layer_offsets[n_layers] == 0) return CELESTE_OK;
    HIP_TRY(hipSetDevice(c->device));
    // vp is uploaded once, stays in HBM across all layers and is written back at the end
    int rc = optim_buffers(c, 1, c->stream);
    if (rc != CELESTE_OK) return rc;
    auto &ob = c->opt;
    const size_t vp_bytes = (size_t)c->S * CEL_P * sizeof(double);
    memcpy(ob.h_vp, vp, vp_bytes);
    HIP_TRY(hipMemcpyAsync(ob.d_vp, ob.h_vp, vp_bytes, hipMemcpyHostToDevice, c->stream));
    rc = joint_run(c, ob.d_vp, nullptr, n_layers, layer_offsets, layer_targets, pos_centers, cfg_in, iterations, f_evals, elbo,
                   status, true);
    // failed targets already hold their pre-layer rows (optim_finalize_kernel); a launch that gave up leaves vp untouched
    if (rc == CELESTE_OK || rc == CELESTE_ERR_NONFINITE_INPUT || rc == CELESTE_ERR_NONFINITE_RESULT) {
        double *const h_out = ob.h_vp + (size_t)c->S * CEL_P;
        if (hipMemcpyAsync(h_out, ob.d_vp, vp_bytes, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipStreamSynchronize(c->stream) != hipSuccess) return CELESTE_ERR_HIP;
        memcpy(vp, h_out, vp_bytes);
    }
    return rc;
} ABI_CATCH

extern "C" int celeste_optim_stats(int reset, uint64_t out[5]) try {
    unsigned long long h[5] = {0, 0, 0, 0, 0};
    if (out) {
        HIP_TRY(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_optim_stats), sizeof h));
        for (int i = 0; i < 5; ++i) out[i] = h[i];
    }
    if (reset) { unsigned long long z[5] = {0, 0, 0, 0, 0}; HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_optim_stats), z, sizeof z)); }
    return CELESTE_OK;
} ABI_CATCH

extern "C" int celeste_tr_solve_batch(int device, int32_t n, const double *H, const double *g, const double *delta,
                                      int32_t solver, int32_t secular_iters, double *p, double *m, int32_t *interior,
                                      int32_t *fell_back) try {
    if (n < 0 || !H || !g || !delta || !p || solver < 0 || solver > 2 || secular_iters < 0) return CELESTE_ERR_INVALID_ARG;
    if (n == 0) return CELESTE_OK;
    int st = select_device(device);
    if (st != CELESTE_OK) return st;
    double *d_H = nullptr, *d_g = nullptr, *d_delta = nullptr, *d_p = nullptr, *d_m = nullptr;
    int32_t *d_i = nullptr, *d_f = nullptr;
    st = dev_upload(&d_H, H, (size_t)n * NF * NF);
    if (st == CELESTE_OK) st = dev_upload(&d_g, g, (size_t)n * NF);
    if (st == CELESTE_OK) st = dev_upload(&d_delta, delta, (size_t)n);
    if (st == CELESTE_OK) st = dev_upload<double>(&d_p, nullptr, (size_t)n * NF);
    if (st == CELESTE_OK) st = dev_upload<double>(&d_m, nullptr, (size_t)n);
    if (st == CELESTE_OK) st = dev_upload<int32_t>(&d_i, nullptr, (size_t)n);
    if (st == CELESTE_OK) st = dev_upload<int32_t>(&d_f, nullptr, (size_t)n);
    if (st == CELESTE_OK) {
        hipLaunchKernelGGL(tr_solve_kernel, dim3((unsigned)n), dim3(64), 0, nullptr, d_H, d_g, d_delta, (int)solver,
                           secular_iters > 0 ? (int)secular_iters : 20, d_p, d_m, d_i, d_f);
        std::vector<int32_t> hi((size_t)n), hf((size_t)n);
        std::vector<double> hm((size_t)n);
        if (hipMemcpy(p, d_p, (size_t)n * NF * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(hm.data(), d_m, (size_t)n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(hi.data(), d_i, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(hf.data(), d_f, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) st = CELESTE_ERR_HIP;
        for (int k = 0; k < n && st == CELESTE_OK; ++k) {
            if (m) m[k] = hm[k];
            if (interior) interior[k] = hi[k];
            if (fell_back) fell_back[k] = hf[k];
        }
    }
    (void)hipFree(d_H); (void)hipFree(d_g); (void)hipFree(d_delta); (void)hipFree(d_p); (void)hipFree(d_m);
    (void)hipFree(d_i); (void)hipFree(d_f);
    return st;
} ABI_CATCH

#ifdef OPTIM_DEBUG_T
extern "C" int celeste_debug_T(int32_t n, double *out) {   // n = 0: allocate for 4096 problems; n > 0: fetch n records
    static double *buf = nullptr;
    if (!buf) { HIP_TRY(hipMalloc((void **)&buf, (size_t)4096 * 4 * NF * sizeof(double))); HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_T), &buf, sizeof buf)); }
    if (n > 0) HIP_TRY(hipMemcpy(out, buf, (size_t)n * 4 * NF * sizeof(double), hipMemcpyDeviceToHost));
    return CELESTE_OK;
}
#endif
#ifdef VALUE_TIMING
extern "C" int celeste_value_clocks(int reset, uint64_t out[8]) {   // debug builds only (tools/variants)
    unsigned long long h[8] = {0};
    HIP_TRY(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_value_clk), sizeof h));
    for (int i = 0; i < 8; ++i) out[i] = h[i];
    if (reset) { unsigned long long z[8] = {0}; HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_value_clk), z, sizeof z)); }
    return CELESTE_OK;
}
#endif
#ifdef LIFT_TIMING
extern "C" int celeste_lift_clocks(int reset, uint64_t out[16]) {   // debug builds only (tools/variants)
    unsigned long long h[16] = {0};
    HIP_TRY(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_lift_clk), sizeof h));
    for (int i = 0; i < 16; ++i) out[i] = h[i];
    if (reset) { unsigned long long z[16] = {0}; HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_lift_clk), z, sizeof z)); }
    return CELESTE_OK;
}
#endif
#ifdef FUSED_TIMING
extern "C" int celeste_fused_clocks(int reset, uint64_t out[16]) {   // debug builds only (tools/variants)
    unsigned long long h[16] = {0};
    HIP_TRY(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_fused_clk), sizeof h));
    for (int i = 0; i < 16; ++i) out[i] = h[i];
    if (reset) { unsigned long long z[16] = {0}; HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_fused_clk), z, sizeof z)); }
    return CELESTE_OK;
}
#endif
#ifdef OPTIM_TIMING
extern "C" int celeste_optim_clocks(int reset, uint64_t out[16]) {   // debug builds only (tools/variants)
    unsigned long long h[16] = {0};
    HIP_TRY(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_optim_clk), sizeof h));
    for (int i = 0; i < 16; ++i) out[i] = h[i];
    if (reset) { unsigned long long z[16] = {0}; HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_optim_clk), z, sizeof z)); }
    return CELESTE_OK;
}
#endif

// ---- expected-image renderer (bin/write_celeste_expectation.jl:112-156, fsm_util.jl:349-400) ------------
extern "C" int celeste_render_expected(celeste_ctx_t *c, const double *vp, int32_t image, double *out_plane) try {
    if (!c || !vp || !out_plane || image < 0 || image >= c->N) return CELESTE_ERR_INVALID_ARG;
    HIP_TRY(hipSetDevice(c->device));
    const DevImage &im = c->imgs->h_images[image];
    const size_t npix = (size_t)im.H * im.W;
    double *d_plane = nullptr;
    if (!c->d_vp) HIP_TRY(hipMalloc((void **)&c->d_vp, (size_t)c->S * CEL_P * sizeof(double)));
    HIP_TRY(hipMemcpyAsync(c->d_vp, vp, (size_t)c->S * CEL_P * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(scratch_get(c, 12, npix * sizeof(double), &d_plane));
    int rc = CELESTE_OK;
    if (hipMemsetAsync(d_plane, 0, npix * sizeof(double), c->stream) != hipSuccess) rc = CELESTE_ERR_HIP;
    if (rc == CELESTE_OK) {
        if (c->V > 0)
            hipLaunchKernelGGL(prep_kernel, dim3((unsigned)c->V), dim3(64), 0, c->stream, c->d_vp, c->d_images,
                               c->d_patches, c->d_vis_src, c->d_vis_img, c->N, c->K, c->d_srcimg, c->d_comps, nullptr,
                               c->d_vis_off, c->M, (int)c->dense, nullptr, nullptr, 0);
        hipLaunchKernelGGL(render_kernel, dim3((unsigned)((size_t)c->S * c->CH)), dim3(64), 0, c->stream, c->d_patches,
                           c->d_coefs, c->d_bitmaps, im.pixels, c->d_srcimg, c->d_comps, (int)image, c->N, c->NC,
                           c->CH, c->chunk_px, im.H, d_plane, c->d_vis_off, c->d_vis_img, (int)c->dense);
        if (hipMemcpyAsync(out_plane, d_plane, npix * sizeof(double), hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = CELESTE_ERR_HIP;
    }
    if (hipStreamSynchronize(c->stream) != hipSuccess) rc = CELESTE_ERR_HIP;
    return rc;
} ABI_CATCH

// ---- one process, N devices (celeste_group_*) ----------------------------------------------------------------------
#include "group.h"
