/*
 * celeste_mi355x.h -- C ABI of the MI355X-native Celeste ELBO engine.
 *
 * Drop-in boundary.  Every entry point below names the
 * reference interface (Celeste.jl, Julia 0.6, paths relative to the reference
 * checkout) that it replaces.  Plain C types only: no torch, no HIP types in
 * the signatures (a stream is passed as void*).
 *
 *   reference                                               this library
 *   ------------------------------------------------------  --------------------------
 *   ElboArgs(images, patches, active_sources)               celeste_ctx_create
 *       src/deterministic_vi/elbo_args.jl:165-211
 *   elbo(ea, vp, elbo_vars, bvn_bundle) -> SensitiveFloat    celeste_elbo_eval
 *       src/deterministic_vi/elbo_objective.jl:482-492      celeste_elbo_eval_batch[_device]
 *   elbo_likelihood(ea, vp, ...)                             same, without CELESTE_FLAG_KL
 *       src/deterministic_vi/elbo_objective.jl:400-474
 *   ImagePatch ctor: stamp conditioning + spline prefilter  celeste_spline_prefilter
 *       src/model/imaged_sources.jl:97-107
 *   PSF.get_psf_at_point / Model.render_psf                  celeste_psf_raster
 *       src/PSF.jl:150-161, src/model/psf_model.jl:61-75
 *   estimate_time (sum of active pixels)                     celeste_ctx_work_stats
 *       src/ParallelRun.jl:45-47
 *   images shared by the per-source ElboArgs of a box        celeste_images_create + celeste_ctx_create_on
 *       src/ParallelRun.jl:468-488 (process_source)
 *   ElboMaximize.maximize!(ea, vp, cfg)                      celeste_maximize_batch[_device]
 *       src/deterministic_vi/ElboMaximize.jl:228-242
 *   one_node_joint_infer's inner loop                         celeste_joint_infer
 *       src/ParallelRun.jl:135-196, 302-397
 *   the N workers of one process draining a source list      celeste_group_create + celeste_group_elbo_eval_batch /
 *       src/ParallelRun.jl:546-607 (one_node_single_infer),      celeste_group_maximize_batch / celeste_group_joint_infer
 *       :302-369 (process_sources_dynamic!), :45-56               (one process, N HIP devices, RCCL inside the library)
 *       (estimate_time, load balancing)
 *
 * Conventions (all taken from the reference):
 *   - matrices are column-major with the first index (h, image row) fastest;
 *     element [h,w] of an H x W image is at  h-1 + H*(w-1)  (1-based h,w);
 *   - pixel coordinates are the real pair (h, w), 1-based;
 *   - a source's variational parameters are 44 doubles in CanonicalParams order
 *     (src/model/param_set.jl:76-107);
 *   - gradient d is 44 doubles, Hessian h is 44 x 44 doubles, column-major,
 *     emitted exactly symmetric.
 */
#ifndef CELESTE_MI355X_H
#define CELESTE_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CELESTE_P 44            /* length(CanonicalParams), param_set.jl:107 */
#define CELESTE_NUM_BANDS 5
#define CELESTE_STAMP 51        /* psfmap stamp edge, AccuracyBenchmark.jl:559 */
#define CELESTE_COEF 53         /* padded B-spline coefficient edge */
#define CELESTE_NUM_COLOR_COMPONENTS 8

/* status codes (replace the reference's @assert / AssertionError).  No C++ exception crosses this ABI: a host allocation
 * that fails inside the library returns CELESTE_ERR_ALLOC, any other host-side failure CELESTE_ERR_HIP. */
enum {
    CELESTE_OK = 0,
    CELESTE_ERR_INVALID_ARG = 1,
    CELESTE_ERR_NONFINITE_INPUT = 2,   /* elbo_objective.jl:487 */
    CELESTE_ERR_NONFINITE_RESULT = 3,  /* elbo_args.jl:145-149 */
    CELESTE_ERR_HIP = 4,
    CELESTE_ERR_NO_DEVICE = 5,
    CELESTE_ERR_ALLOC = 6,
    CELESTE_ERR_ABORTED = 7            /* celeste_group_*: the group's communicators were torn down (see "Errors" there) */
};

/* evaluation flags: has_gradient / has_hessian of the result SensitiveFloat
 * (SensitiveFloats.jl:37-47; has_hessian implies has_gradient) and
 * ElboArgs.include_kl (elbo_args.jl:189). */
enum {
    CELESTE_FLAG_GRAD = 1u,
    CELESTE_FLAG_HESS = 2u,
    CELESTE_FLAG_KL = 4u,
    /* single-precision pixel arithmetic (BASELINE config 5, tolerance 1e-4 against the fp64 result): the galaxy
     * component loop (packed, two components per instruction), the star spline, the per-pixel term and the record
     * entries are fp32, and so are the star / galaxy densities of the neighbours' pre-rendered light (their moments E,
     * var are formed in fp64); the chunk records everything is summed into, the lift and the KL stay fp64.  Measured against the
     * fp64 path on all 30 000 sources of config 5: 6e-6 / 5e-7 / 1.2e-6 on v / d / h.  No reference counterpart. */
    CELESTE_FLAG_FP32 = 8u,
    /* split variant of the pixel sum (measurement aid): the pixel kernel writes one
     * 68-double record per visited pixel to HBM and a separate streaming kernel forms the per-patch sums
     * (the accumulation of add_pixel_term! into elbo_vars.elbo, elbo_objective.jl:330-392,452-466).
     * Requires CELESTE_FLAG_HESS; distinct targets in the batch.  Same results up to summation order. */
    CELESTE_FLAG_SPLIT = 16u,
    /* Packed Hessian output: h holds CELESTE_HP = 990 doubles per target instead of 44 x 44 -- the upper triangle,
     * column by column: element (i, j), i <= j, at j (j + 1) / 2 + i.  Halves the D2H traffic of the host-pointer
     * entry points; the consumer (propagate_derivatives!, ConstraintTransforms.jl:379-393) symmetrises anyway
     * (:452-457).  Not valid with celeste_elbo_eval_multi. */
    CELESTE_FLAG_PACKED_HESS = 32u
};
#define CELESTE_HP 990          /* 44 * 45 / 2 */

/* Model.Image (src/model/image_model.jl:6-38).  Borrowed for the duration of
 * celeste_ctx_create only. */
typedef struct celeste_image_t {
    int32_t H;                    /* rows */
    int32_t W;                    /* columns */
    int32_t band;                 /* 1..5 (u,g,r,i,z) */
    int32_t reserved;
    const float *pixels;          /* H*W electrons; NaN = masked */
    const float *sky;             /* H*W nanomaggies (img.sky[h,w]) */
    const float *nelec_per_nmgy;  /* H, one per row */
} celeste_image_t;

/* Model.ImagePatch (src/model/imaged_sources.jl:60-71). */
typedef struct celeste_patch_t {
    int32_t off_h;                /* bitmap_offset[1] = first(box[1]) - 1 */
    int32_t off_w;                /* bitmap_offset[2] */
    int32_t H2;                   /* size(active_pixel_bitmap, 1); may be 0 */
    int32_t W2;
    const uint8_t *bitmap;        /* H2*W2 column-major; NULL => !isnan(pixel) */
    double wcs_jacobian[4];       /* 2x2 column-major */
    double world_center[2];
    double pixel_center[2];
    const double *psf;            /* psf_K x {alphaBar, xiBar1, xiBar2, tauBar11, tauBar12, tauBar22} */
    int32_t stamp;                /* index into celeste_problem_t.stamps */
    int32_t reserved;
} celeste_patch_t;

/* Model.PriorParams (src/model/light_source_model.jl:78-132).  index 0 = star,
 * 1 = galaxy. */
typedef struct celeste_prior_t {
    double is_star[2];
    double flux_mean[2];
    double flux_var[2];
    double k[2][8];               /* prior.k[:, i] */
    double color_mean[2][8][4];   /* prior.color_mean[:, d, i] */
    double color_cov[2][8][16];   /* prior.color_cov[:, :, d, i], column-major 4x4 */
    double gal_radius_px_mean;
    double gal_radius_px_var;
} celeste_prior_t;

/* The whole-field problem: all images, every catalogued source's patches and
 * the neighbour graph (Model.find_neighbors, imaged_sources.jl:232-244).
 * Evaluating target t is the reference's
 *     ElboArgs(images, patches[[t; neighbors(t)], :], [1])
 * (ParallelRun.jl:468-488). */
typedef struct celeste_problem_t {
    int32_t n_images;
    int32_t n_sources;
    int32_t psf_K;                /* ElboArgs.psf_K, default 2 */
    int32_t n_stamps;
    const celeste_image_t *images;      /* n_images */
    const celeste_patch_t *patches;     /* [s * n_images + n], or the sparse list described below */
    const double *stamps;               /* n_stamps x 51 x 51, raw psfmap(...) output */
    const int64_t *nbr_offsets;         /* n_sources + 1 (CSR) */
    const int32_t *nbr_index;           /* 0-based source ids */
    const celeste_prior_t *prior;       /* NULL => built-in cfg/{star,gal}_prior tables */
    /* Sparse patch table for many-image problems (overlapping fields: a source has a non-empty patch in the few
     * images that cover it and the reference's empty clamp_box in all others, imaged_sources.jl:10-14).  When
     * n_patch_entries > 0, `patches` holds only those n_patch_entries entries, sorted by (source, image), entry k
     * being the patch of source patch_source[k] in image patch_image[k]; every pair not listed has an empty
     * patch.  n_patch_entries == 0 selects the dense [s * n_images + n] table above. */
    int64_t n_patch_entries;
    const int32_t *patch_source;
    const int32_t *patch_image;
} celeste_problem_t;

typedef struct celeste_ctx celeste_ctx_t;
typedef struct celeste_images celeste_images_t;

/* Per-sweep work statistics. */
typedef struct celeste_work_stats_t {
    int64_t n_targets;
    int64_t active_pixel_visits;    /* sum over targets, images of visited pixels */
    int64_t patch_rows;             /* sum of H2 */
    int64_t neighbor_links;         /* sum of K_s */
    int64_t algorithmic_bytes;      /* 9 A + 4 R + 352 (1+K) + 200 per non-empty patch of the target and its neighbours + 8288, per target */
    int64_t record_bytes;           /* split variant: 544 A + 544 per non-empty (target, image) patch */
    int64_t record_tiles;           /* split variant: 64-pixel record tiles actually stored / re-read */
} celeste_work_stats_t;

/* ABI version: major * 100 + minor.  200: celeste_optim_config_t carries tr_secular_iters (round 2 grew the struct by 8
 * bytes without bumping the version); celeste_maximize_batch_device and celeste_joint_infer exist.  Bindings check it
 * (cabi.load_library, shim/CelesteMI355X.jl): a caller built against another major version must not pass structs.
 * 210: the celeste_group_* entry points (one process, N devices).
 * 220: CELESTE_ERR_ABORTED, celeste_group_collectives; celeste_group_joint_infer exchanges once per SEGMENT of batches. */
#define CELESTE_ABI_VERSION 220
int celeste_version(void);
const char *celeste_strerror(int status);

/* Uploads images, patch descriptors, conditioned spline coefficients and the
 * neighbour graph to HBM of `device`.  Fails with CELESTE_ERR_NO_DEVICE when no
 * HIP device is present: there is no CPU fallback. */
int celeste_ctx_create(const celeste_problem_t *problem, int device, celeste_ctx_t **out);
/* Waits for the context's own work, releases its device and page-locked memory.  Its two HIP streams are NOT destroyed: they
 * go to a per-device pool of idle streams (at most CELESTE_STREAM_POOL_MAX, default 32; an idle stream holds about 1 MB of
 * device memory) from which the next context of the process takes them -- a caller that creates a context per source
 * (ParallelRun.jl:468-488) neither pays for stream creation each time nor meets hipStreamDestroy, which the HIP runtime of
 * ROCm 7.0 does not survive cleanly under host load (profiles/r08_stale_stream_write.md).  Its page-locked staging blocks of up
 * to 1 MB are kept for the next context as well (CELESTE_PINNED_POOL_KB, default 16 MB per process): creating a per-source
 * context on an image handle, one evaluation and destroying it take 0.17 + 0.09 + 0.02 ms (1.1 ms with every block locked
 * and released each time and every table copied synchronously; tools/gpu_per_source_ctx_time.py). */
void celeste_ctx_destroy(celeste_ctx_t *ctx);

/* Shared image handle.  The reference builds one ElboArgs per source over the SAME images
 * (process_source, ParallelRun.jl:468-488: `ElboArgs(images, patches[[t; neighbors], :], [1])`); uploading the
 * planes (and computing their lgamma / log-iota planes) once and creating every per-source context on the handle
 * makes such a context cost a patch-table upload.  celeste_images_create copies the pixel / sky / calibration
 * planes of `images` to HBM of `device`; celeste_ctx_create_on is celeste_ctx_create with problem->images ignored
 * (problem->n_images must equal the handle's; patch boxes are validated against the handle's image sizes).  The
 * handle is reference counted: celeste_images_destroy drops the creator's reference, device memory is released
 * when the last context created on it is destroyed as well.  celeste_ctx_create is the two calls in sequence. */
int celeste_images_create(int32_t n_images, const celeste_image_t *images, int device, celeste_images_t **out);
void celeste_images_destroy(celeste_images_t *images);
int celeste_ctx_create_on(celeste_images_t *images, const celeste_problem_t *problem, celeste_ctx_t **out);

/* Concurrency contract (all entry points that take a ctx): a context owns its scratch tables (per-source constants,
 * pre-rendered neighbour light, work lists, pixel-sum records), so AT MOST ONE call per context may be in flight at
 * any time -- including asynchronous celeste_elbo_eval_batch_device launches that have not completed on their
 * stream.  Callers that want concurrency use one context per thread / stream (cheap on a shared image handle);
 * contexts never synchronise each other: the host-pointer entry points run on a private non-blocking stream of the
 * context and wait for that stream only. */

/* Page-locked host memory for the host-pointer entry points.  A Hessian array that lives in memory obtained from
 * celeste_host_alloc, or registered with celeste_host_register, is written by the kernels themselves (its device address,
 * hipHostGetDevicePointer): one launch chain, no copy; other outputs of such memory are copied by DMA.  Pageable buffers
 * work too: large batches are then cut into parts whose copies (through page-locked staging) overlap the kernels of the next
 * part.  (hipHostMalloc / hipHostRegister; no reference counterpart.) */
void *celeste_host_alloc(size_t bytes);
void celeste_host_free(void *ptr);
int celeste_host_register(void *ptr, size_t bytes);
int celeste_host_unregister(void *ptr);

/* elbo(ea, vp) for one target (Sa = 1, neighbours value-only).
 * vp: n_sources x 44 host doubles (row s = source s).  Outputs may be NULL when
 * the corresponding flag is off.  Counters: elbo_args.jl:62-63. */
int celeste_elbo_eval(celeste_ctx_t *ctx, const double *vp, int32_t target, uint32_t flags,
                      double *v, double *d, double *h,
                      int64_t *n_active_px, int64_t *n_inactive_px);

/* One launch for a whole batch of targets that may be evaluated together
 * (a Cyclades batch, or every source of the field for an evaluate-only sweep).
 * Host pointers; v[n], d[n*44], h[n*44*44] (n*990 with CELESTE_FLAG_PACKED_HESS), counters[n*2], status[n].
 * Page-locked Hessian arrays are filled by the kernels directly; with pageable ones large batches are cut into parts whose
 * device-to-host copies overlap the kernels of the next part (results do not depend on the cut).  Targets may repeat.  Returns the first non-OK per-target status, if any; the outputs of
 * the other targets are valid. */
int celeste_elbo_eval_batch(celeste_ctx_t *ctx, const double *vp, int32_t n_targets,
                            const int32_t *targets, uint32_t flags,
                            double *v, double *d, double *h,
                            int64_t *counters, int32_t *status);

/* elbo() with several active sources (ElboArgs.active_sources with Sa > 1, elbo_args.jl:165-211; the layout of
 * SensitiveFloats.jl:29-31): d is P x Sa column-major (column a = active[a]), h is (P Sa) x (P Sa) column-major
 * with the cross blocks of add_sources_sf! / combine_sfs_hessian!, exactly symmetric.  Every other source of the
 * context that is a neighbour of an active source contributes value-only.  A pixel of several active patches is
 * visited once (elbo_objective.jl:430-470).  Production always uses Sa = 1 (ParallelRun.jl:482): this entry point
 * serves the reference's multi-source tests and is not tuned. */
int celeste_elbo_eval_multi(celeste_ctx_t *ctx, const double *vp, int32_t n_active, const int32_t *active,
                            uint32_t flags, double *v, double *d, double *h,
                            int64_t *n_active_px, int64_t *n_inactive_px);

/* Same, all pointers already in HBM, asynchronous on `stream` (a hipStream_t,
 * NULL = default stream).  d_status[n] receives per-target status codes. */
int celeste_elbo_eval_batch_device(celeste_ctx_t *ctx, const double *d_vp, int32_t n_targets,
                                   const int32_t *d_targets, uint32_t flags,
                                   double *d_v, double *d_d, double *d_h,
                                   int64_t *d_counters, int32_t *d_status, void *stream);

/* HIP-event timing of the kernels of the most recent batch launch, on the
 * stream they were launched on.  Enable before the launch; read after the
 * stream has been synchronised.  ms[0] = per-source preparation kernel,
 * ms[1] = pixel kernel, ms[2] = 44-space lift kernel. */
int celeste_ctx_enable_timing(celeste_ctx_t *ctx, int enable);
int celeste_ctx_last_kernel_ms(celeste_ctx_t *ctx, float ms[3]);
/* CELESTE_FLAG_SPLIT launches only: duration of the record-sum kernel (ms[1] above is then the
 * record-writing pixel kernel alone, ms[2] the lift alone). */
int celeste_ctx_last_record_sum_ms(celeste_ctx_t *ctx, float *ms);

int celeste_ctx_work_stats(celeste_ctx_t *ctx, int32_t n_targets, const int32_t *targets,
                           celeste_work_stats_t *out);

/* ImagePatch ctor arithmetic: max(.,0), +1e-6, normalise, softpluslike, then the
 * cubic B-spline prefilter with Line() boundaries on a padded 53x53 grid -- one stamp, on the host (a utility for
 * callers and tests: celeste_ctx_create does the same for every stamp of the problem on the device, same
 * operations in the same order, coefficients equal to these to the last bit or two of the logarithm). */
int celeste_spline_prefilter(const double *stamp51, double *coef53);
/* The coefficients the context holds for stamp `stamp` (53 x 53, as celeste_spline_prefilter lays them out), copied down
 * from the device: a diagnostic / test accessor for the device-side constructor arithmetic. */
int celeste_ctx_spline_coefficients(celeste_ctx_t *ctx, int32_t stamp, double *coef53);

/* Gaussian-mixture PSF raster sum_k alphaBar_k N(x; xiBar_k, tauBar_k) on the
 * grid rows x cols (get_psf_at_point; render_psf uses rows = cols = -25:25).
 * psf: K x 6 as in celeste_patch_t.  out: n_rows x n_cols column-major, host. */
int celeste_psf_raster(int device, const double *psf, int32_t K,
                       const double *rows, int32_t n_rows,
                       const double *cols, int32_t n_cols, double *out);

/* ---- the caller of the hot path ------------------------------------------
 * ElboMaximize.ElboConfig defaults (src/deterministic_vi/ElboMaximize.jl:43-49, 95-108). */
typedef struct celeste_optim_config_t {
    double loc_width;      /* half-width of the position box around the initial position, 1e-4 */
    double loc_scale;      /* 1.0 */
    int32_t max_iters;     /* 50 */
    int32_t include_kl;    /* ElboArgs.include_kl, 1 */
    double xtol_abs;       /* 1e-7 */
    double ftol_rel;       /* 1e-6 */
    double gtol;           /* 1e-8 */
    double initial_delta;  /* 1.0 */
    double delta_hat;      /* 1e9 */
    int32_t tr_secular_iters;  /* cap of the Newton iterations on the trust-region multiplier: 0 = run to convergence
                                * (<= 20); 5 = Optim.jl's solve_tr_subproblem! default (third-party, stops after 5
                                * whether or not converged) for comparisons with a real Optim.jl run */
    int32_t reserved;
} celeste_optim_config_t;

/* ElboMaximize.maximize!(ea, vp, cfg) (ElboMaximize.jl:228-242) for a batch of targets, entirely on the device:
 * enforce! / to_free! (ConstraintTransforms.jl:84-126, 225-253), then Newton trust-region iterations on the 41
 * free parameters of every target in lock-step (one elbo() per iteration and target, propagate_derivatives!
 * analytically, exact trust-region sub-problem in the eigenbasis), to_bound! at the end.  Every target sees its
 * neighbours frozen at the input `vp` (ParallelRun.process_source, ParallelRun.jl:468-498); for a conflict-free
 * Cyclades batch that is also the joint-inference semantics (ParallelRun.jl:372-397).  vp (n_sources x 44, host)
 * is updated in place for the targets only.  vp_neighbors (n_sources x 44, may be NULL = vp) holds the frozen
 * parameters under which every source acts as a *neighbour* (single inference starts targets at
 * generic_init_source while their neighbours sit at catalog_init_source, DeterministicVI.jl:94-103).
 * pos_centers (n_targets x 2, may be NULL = current position) are the centres of the position boxes, which the
 * reference keeps fixed across repeated maximize! calls (ParallelRun.jl:96-100).  cfg == NULL selects the
 * defaults above.  Per-target outputs may be NULL.  Targets must be distinct (CELESTE_ERR_INVALID_ARG otherwise: two
 * optimisations of one source would share its row of vp).  A target whose ELBO becomes non-finite stops with its
 * status set (CELESTE_ERR_NONFINITE_*), its row of vp is left as it was on entry, and every other target is optimised
 * normally -- the reference catches per source and keeps the rest (ParallelRun.jl:582-597, 389-396); the return
 * value is the first such status. */
int celeste_maximize_batch(celeste_ctx_t *ctx, double *vp, const double *vp_neighbors, const double *pos_centers,
                           int32_t n_targets, const int32_t *targets, const celeste_optim_config_t *cfg,
                           int32_t *iterations, int32_t *f_evals, double *elbo, int32_t *status);

/* celeste_maximize_batch with every pointer already in HBM, on `stream` (a hipStream_t, NULL = default stream): d_vp
 * (n_sources x 44) is optimised in place for the targets -- the parameter table never leaves the device, which is what
 * lets a multi-GPU driver all-gather the optimised rows over RCCL straight from it (parallel.sharded_maximize).
 * d_vp_neighbors (may be NULL = d_vp) and d_pos_centers (n_targets x 2, may be NULL) as above; the per-target outputs
 * d_iterations / d_f_evals / d_elbo / d_status (device, may be NULL) are written when the batch is done.  Targets must be
 * distinct (not checked here: they live on the device).  Batches of up to 640 targets run as ONE persistent launch
 * (every target iterates at its own pace) and the call is asynchronous; larger batches run the lock-step driver,
 * which blocks the host until the batch has converged.  A failing target gets its input row back and its status set
 * (CELESTE_ERR_HIP for every target if the launch itself gave up). */
int celeste_maximize_batch_device(celeste_ctx_t *ctx, double *d_vp, const double *d_vp_neighbors,
                                  const double *d_pos_centers, int32_t n_targets, const int32_t *d_targets,
                                  const celeste_optim_config_t *cfg, int32_t *d_iterations, int32_t *d_f_evals,
                                  double *d_elbo, int32_t *d_status, void *stream);

/* ParallelRun.one_node_joint_infer's inner loop (ParallelRun.jl:135-196, 302-397) for a schedule the caller has laid
 * out: `n_layers` layers, layer l = layer_targets[layer_offsets[l] .. layer_offsets[l + 1]).  The sources of a layer are
 * optimised simultaneously (celeste_maximize_batch semantics: neighbours frozen), so no two of them may be neighbours
 * (CELESTE_ERR_INVALID_ARG otherwise) -- the j-th sources of the connected components of a Cyclades batch
 * (partition.jl:173-236) are such a layer; every layer sees the parameter table as the layers before it left it, which
 * reproduces the reference's sequential-within-component schedule.  Repeat the layers for num_joint_vi_iters sweeps.
 * vp (n_sources x 44, host) is uploaded once, stays in HBM across all layers and is written back at the end.
 * pos_centers (2 doubles per entry of layer_targets, may be NULL = the position at the start of that layer): the
 * centres of the position boxes, which the reference pins at the initial positions (ParallelRun.jl:96-100).  The
 * per-entry outputs (may be NULL) are indexed like layer_targets.  A source that fails in some layer keeps the row it
 * had before that layer (status set, the rest of the schedule goes on: ParallelRun.jl:389-396); the return value is the
 * first such status.
 * Execution: schedules whose layers hold up to 1024 sources run as ONE launch in which an optimisation starts as soon as
 * the optimisations it depends on -- the earlier ones of its source and of its neighbours -- have ended (a dataflow over
 * the entries; the layer boundaries themselves are not waited for); wider schedules run layer by layer.  Both leave the
 * same table, the same per-entry outputs, bit for bit (CELESTE_JOINT_DATAFLOW=0 / 1 forces one or the other). */
int celeste_joint_infer(celeste_ctx_t *ctx, double *vp, int32_t n_layers, const int64_t *layer_offsets,
                        const int32_t *layer_targets, const double *pos_centers, const celeste_optim_config_t *cfg,
                        int32_t *iterations, int32_t *f_evals, double *elbo, int32_t *status);

/* Diagnostics of the trust-region sub-problems solved since the last reset, summed over all contexts of the
 * process: out[0] interior Newton steps, out[1] boundary solutions, out[2] hard cases, out[3] total and out[4]
 * maximum number of secular-equation iterations. */
int celeste_optim_stats(int reset, uint64_t out[5]);

/* The trust-region sub-problem of the optimiser on its own (diagnostic / test entry; Optim.jl's solve_tr_subproblem!,
 * third-party and unvendored, restated from N&W section 4.3): for each of n problems minimise g'p + p'Hp/2 subject to
 * |p| <= delta over the CELESTE_NF = 41 free parameters.  H: n x 41 x 41 (symmetric, either order), g: n x 41,
 * delta: n; out p: n x 41, m (may be NULL): model value, interior (may be NULL): 1 = plain Newton step inside the
 * region, fell_back (may be NULL): 1 = the tridiagonal-space solver handed the problem to the eigen-decomposition.
 * solver 0: as celeste_maximize_batch (tridiagonal space, eigen-decomposition for clusters of > 4 lowest
 * eigenvalues), 1: eigen-decomposition, 2: tridiagonal space only (p = 0 where fell_back).  secular_iters 0 = to
 * convergence (celeste_optim_config_t.tr_secular_iters).  Host pointers. */
#define CELESTE_NF 41
int celeste_tr_solve_batch(int device, int32_t n, const double *H, const double *g, const double *delta,
                           int32_t solver, int32_t secular_iters, double *p, double *m, int32_t *interior,
                           int32_t *fell_back);

/* Expected light of all sources on image `image` (0-based): out[h,w] = sum_s E_G_s.v in nanomaggies, i.e.
 * elbo_vars.E_G.v - sky of the value-only add_pixel_term! sweep in bin/write_celeste_expectation.jl:112-156
 * (every source contributes on its own patch, last column and inactive pixels excluded).  out: H x W doubles,
 * column-major, host. */
int celeste_render_expected(celeste_ctx_t *ctx, const double *vp, int32_t image, double *out_plane);

/* ---- one process, N devices: the source-partition loop --------------------------
 * The reference drains one source list with N workers inside ONE process (one_node_single_infer, ParallelRun.jl:546-607;
 * process_sources_dynamic!, ParallelRun.jl:302-369; its Julia caller is the only caller it has).  A group is that loop over
 * the HIP devices of a node: the images are replicated on every member device, the targets of a call are sharded over the
 * members by cost (estimate_time = the pixels of a source's patches, ParallelRun.jl:45-56; longest first onto the least
 * loaded member, ties by index -- deterministic), every member runs its shard on its own worker thread and stream, and the
 * per-source results are exchanged with ONE ncclAllGather (RCCL over xGMI; communicators from ncclCommInitAll) -- the
 * catalog gather, the only exchange of the path.  Results are those of the one-device entry points, bit for bit: a
 * target's evaluation / optimisation does not depend on what else is in its launch.
 *
 * devices[n_members]: HIP device ordinals (NULL = 0 .. n_members - 1).  Distinct devices exchange over RCCL -- a group of
 * one included (the collective then runs with one rank).  A device that appears more than once makes the group exchange
 * by plain device-to-device copies instead (RCCL refuses duplicate devices in a communicator): two members on one GPU
 * exercise the shard / gather bookkeeping on a one-GPU box; it is a test configuration, not a fast one.
 * At most 16 members.  One call per group at a time: every entry point takes the group's lock for its whole duration (a
 * second thread's call waits); celeste_group_sweep returns with its sweep in flight and the lock released.
 * CELESTE_GROUP_EXCHANGE=rccl / peer forces an exchange mode (tests; real RCCL refuses repeated devices).
 * Errors: per-source failures are statuses, as in the one-device entry points.  A member whose LAUNCH fails still enqueues
 * every collective of the call (the others' results are sound) and the call returns its error.  A member that fails in front
 * of a row exchange's host barrier wakes the barrier: every member leaves the call with an error, nobody enters the
 * collective.  A member that leaves a call WITHOUT a collective the others enqueue (a HIP call failed between the barrier and
 * the collective, or in front of a sweep's gather -- the sweep path has no host barrier, its gathers overlap the next sweep)
 * is detected by count -- every member counts the collectives it has enqueued, the dispatcher knows how many the call
 * holds -- and answered by ncclCommAbort on every communicator, which releases the streams that wait inside the collective:
 * the call (celeste_group_sweep_wait for sweeps in flight) returns CELESTE_ERR_ABORTED, and so does every later call until
 * the group is destroyed and created again.  The same happens when ncclCommGetAsyncError reports an error while a stream
 * holding a collective is waited for, or when that wait exceeds CELESTE_GROUP_TIMEOUT_MS (default 300 000; 0 = no limit).
 * (The reference logs a failing source and goes on, ParallelRun.jl:389-396: that is the per-source status; a device that
 * drops out has no counterpart there.) */
typedef struct celeste_group celeste_group_t;
enum { CELESTE_EXCHANGE_RCCL = 1, CELESTE_EXCHANGE_PEER_COPY = 2 };
typedef struct celeste_group_info_t {
    int32_t n_members;
    int32_t n_devices;      /* distinct devices */
    int32_t exchange;       /* CELESTE_EXCHANGE_* */
    int32_t rccl_ranks;     /* ncclCommCount of the group's communicator (0 without RCCL) */
    int32_t devices[16];
} celeste_group_info_t;
int celeste_group_create(const celeste_problem_t *problem, int32_t n_members, const int32_t *devices, celeste_group_t **out);
void celeste_group_destroy(celeste_group_t *group);
int celeste_group_info(celeste_group_t *group, celeste_group_info_t *out);
/* enqueued[n_members] (may be NULL): the collectives -- catalog gathers, row exchanges -- each member has enqueued on its
 * communicator since the group was created (equal on all members whenever no call is in flight); *aborted (may be NULL): 1 once
 * the communicators have been torn down. */
int celeste_group_collectives(celeste_group_t *group, int64_t *enqueued, int32_t *aborted);

/* celeste_elbo_eval_batch over the members: same arguments, same outputs in the caller's order.  v / d / counters / status
 * of every target are all-gathered to every member (member 0 hands them to the host); Hessians come down from the member
 * that owns the target, all members at once.  CELESTE_FLAG_SPLIT is not supported here. */
int celeste_group_elbo_eval_batch(celeste_group_t *group, const double *vp, int32_t n_targets, const int32_t *targets,
                                  uint32_t flags, double *v, double *d, double *h, int64_t *counters, int32_t *status);

/* The same sweep with everything resident in HBM (what an optimiser loop or a benchmark repeats): _plan shards the targets
 * and uploads the table and the shards; _sweep starts one evaluation of every shard + its catalog gather and returns (the
 * gather of sweep k overlaps the kernels of sweep k + 1 on a second stream; two gather blocks alternate); _wait returns when
 * every started sweep and gather is complete; _results hands out the last sweep (any pointer may be NULL). */
int celeste_group_sweep_plan(celeste_group_t *group, const double *vp, int32_t n_targets, const int32_t *targets, uint32_t flags);
int celeste_group_sweep(celeste_group_t *group);
int celeste_group_sweep_wait(celeste_group_t *group);
int celeste_group_sweep_results(celeste_group_t *group, double *v, double *d, double *h, int64_t *counters, int32_t *status);
/* sizes[n_members] / costs[n_members] (either may be NULL) of the planned shards */
int celeste_group_shard_sizes(celeste_group_t *group, int32_t *sizes, int64_t *costs);
/* HIP-event durations of the last sweep per member: eval_ms[r] = its launch chain, gather_ms[r] = from the end of the chain to
 * the end of its catalog gather.  Enable before the sweep; read after _wait. */
int celeste_group_enable_timing(celeste_group_t *group, int enable);
int celeste_group_last_sweep_ms(celeste_group_t *group, float *eval_ms, float *gather_ms);
/* celeste_ctx_last_kernel_ms of one member's last launch chain: ms[0] preparation, ms[1] pixel kernel, ms[2] lift */
int celeste_group_last_kernel_ms(celeste_group_t *group, int32_t member, float ms[3]);

/* celeste_maximize_batch over the members (one_node_single_infer, ParallelRun.jl:546-607): every member optimises its shard
 * against its copy of the table (neighbours frozen, so shards are independent), then the optimised rows + per-target outputs
 * are all-gathered and every member's table brought up to date.  Arguments and outputs as celeste_maximize_batch. */
int celeste_group_maximize_batch(celeste_group_t *group, double *vp, const double *vp_neighbors, const double *pos_centers,
                                 int32_t n_targets, const int32_t *targets, const celeste_optim_config_t *cfg,
                                 int32_t *iterations, int32_t *f_evals, double *elbo, int32_t *status);

/* one_node_joint_infer (ParallelRun.jl:135-196, 302-397) over the members.  The schedule is given as the reference has it:
 * n_batches Cyclades batches (partition.jl:173-236), batch b = the connected components batch_offsets[b] .. batch_offsets[b+1],
 * component k = the sources comp_targets[comp_offsets[k] .. comp_offsets[k+1]) in the order they are optimised.  Components of
 * a batch never conflict (checked: CELESTE_ERR_INVALID_ARG if a source appears twice in a batch or has a neighbour in another
 * component of it), so they are sharded over the members by cost; a member optimises the sources of its components one
 * after another against its table (celeste_joint_infer's schedule).  A member's table needs the others' rows only when it is
 * about to READ one -- a target's own row, a neighbour's -- that another member has written since the last exchange; the shards
 * and the neighbour graph are known on the host, so the (sweep, batch) steps are cut into SEGMENTS, maximal runs in front of
 * which no such read exists: one launch chain per member and ONE exchange per segment.  A group of one runs the whole call as
 * celeste_joint_infer's single launch + one exchange; a crowded field on several members exchanges once per batch -- at most
 * n_batches exchanges per sweep, never one per layer.  The batches are repeated n_sweeps times (Config.num_joint_vi_iters).
 * pos_centers: 2 doubles per entry of comp_targets (may be NULL), the same in every sweep (ParallelRun.jl:96-100).  Per-entry
 * outputs (may be NULL): [sweep * n_entries + entry].  *n_exchanges (may be NULL) receives the number of exchanges made.
 * Equals celeste_joint_infer on the flattened schedule bit for bit. */
int celeste_group_joint_infer(celeste_group_t *group, double *vp, int32_t n_sweeps, int32_t n_batches,
                              const int64_t *batch_offsets, const int64_t *comp_offsets, const int32_t *comp_targets,
                              const double *pos_centers, const celeste_optim_config_t *cfg, int32_t *iterations,
                              int32_t *f_evals, double *elbo, int32_t *status, int64_t *n_exchanges);

#ifdef __cplusplus
}
#endif
#endif /* CELESTE_MI355X_H */
