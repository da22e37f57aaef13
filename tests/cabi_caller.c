/*
 * cabi_caller.c -- a plain C program compiled against include/celeste_mi355x.h and nothing else of this repository.
 *
 * This is the caller the reference's `ccall` would be (src/deterministic_vi/ElboMaximize.jl:166,
 * `bound_result = elbo(ea, vp, get_elbo_vars(), cfg.bvn_bundle)`, and the joint-inference loop of
 * src/ParallelRun.jl:302-397): it fills celeste_problem_t / celeste_image_t / celeste_patch_t the way the header
 * documents them, with the compiler -- not a ctypes mirror -- laying out the structs, and calls
 *     celeste_ctx_create, celeste_elbo_eval (one call per source), celeste_elbo_eval_batch,
 *     celeste_maximize_batch, celeste_joint_infer, celeste_ctx_destroy,
 * and the same three computations through a device group (one process, N devices: the N workers of
 * src/ParallelRun.jl:546-607): celeste_group_create, celeste_group_elbo_eval_batch, celeste_group_maximize_batch,
 * celeste_group_joint_infer, celeste_group_destroy.
 * Input: a fixture of tests/golden/export_raw.py (tests/golden/raw/<name>.txt manifest + <name>.bin, little-endian,
 * matrices column-major).  Output on stdout, one record per line, every double as %.17g:
 *     elbo <s> <v> <n_active_px> <n_inactive_px>
 *     d <s> <44 numbers>
 *     h <s> <44 x 44 numbers, column-major>
 *     batch_equal <0|1>                      (celeste_elbo_eval_batch returned the single calls' numbers, bit for bit)
 *     maximize <s> <iterations> <f_evals> <status> <elbo> <44 optimised parameters>
 *     joint <entry> <source> <iterations> <f_evals> <status> <elbo>
 *     joint_vp <s> <44 parameters>
 *     group <n_members> <exchange> <rccl_ranks>
 *     group_equal <eval 0|1> <maximize 0|1> <joint 0|1> <exchanges>
 *                                            (celeste_group_* over `members` members on `device` returned the one-device
 *                                            calls' numbers, bit for bit; one exchange per batch and sweep)
 * tests/test_cabi_caller.py builds it with `gcc -std=c99`, runs it on the GPU box and compares with the committed
 * golden (tests/golden/<name>.npz) at the 1e-8 of BASELINE.json, and with the ctypes binding's results bit for bit.
 *
 * The _Static_assert lines pin the struct layout the header promises on the LP64 ABI; tests/test_cabi.py parses them
 * and holds the ctypes mirrors (celeste.jl_amd/cabi.py) to the same offsets.
 */
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "celeste_mi355x.h"

_Static_assert(sizeof(celeste_image_t) == 40, "celeste_image_t");
_Static_assert(offsetof(celeste_image_t, H) == 0, "celeste_image_t.H");
_Static_assert(offsetof(celeste_image_t, W) == 4, "celeste_image_t.W");
_Static_assert(offsetof(celeste_image_t, band) == 8, "celeste_image_t.band");
_Static_assert(offsetof(celeste_image_t, pixels) == 16, "celeste_image_t.pixels");
_Static_assert(offsetof(celeste_image_t, sky) == 24, "celeste_image_t.sky");
_Static_assert(offsetof(celeste_image_t, nelec_per_nmgy) == 32, "celeste_image_t.nelec_per_nmgy");
_Static_assert(sizeof(celeste_patch_t) == 104, "celeste_patch_t");
_Static_assert(offsetof(celeste_patch_t, off_h) == 0, "celeste_patch_t.off_h");
_Static_assert(offsetof(celeste_patch_t, off_w) == 4, "celeste_patch_t.off_w");
_Static_assert(offsetof(celeste_patch_t, H2) == 8, "celeste_patch_t.H2");
_Static_assert(offsetof(celeste_patch_t, W2) == 12, "celeste_patch_t.W2");
_Static_assert(offsetof(celeste_patch_t, bitmap) == 16, "celeste_patch_t.bitmap");
_Static_assert(offsetof(celeste_patch_t, wcs_jacobian) == 24, "celeste_patch_t.wcs_jacobian");
_Static_assert(offsetof(celeste_patch_t, world_center) == 56, "celeste_patch_t.world_center");
_Static_assert(offsetof(celeste_patch_t, pixel_center) == 72, "celeste_patch_t.pixel_center");
_Static_assert(offsetof(celeste_patch_t, psf) == 88, "celeste_patch_t.psf");
_Static_assert(offsetof(celeste_patch_t, stamp) == 96, "celeste_patch_t.stamp");
_Static_assert(sizeof(celeste_prior_t) == 2752, "celeste_prior_t");
_Static_assert(offsetof(celeste_prior_t, is_star) == 0, "celeste_prior_t.is_star");
_Static_assert(offsetof(celeste_prior_t, flux_mean) == 16, "celeste_prior_t.flux_mean");
_Static_assert(offsetof(celeste_prior_t, flux_var) == 32, "celeste_prior_t.flux_var");
_Static_assert(offsetof(celeste_prior_t, k) == 48, "celeste_prior_t.k");
_Static_assert(offsetof(celeste_prior_t, color_mean) == 176, "celeste_prior_t.color_mean");
_Static_assert(offsetof(celeste_prior_t, color_cov) == 688, "celeste_prior_t.color_cov");
_Static_assert(offsetof(celeste_prior_t, gal_radius_px_mean) == 2736, "celeste_prior_t.gal_radius_px_mean");
_Static_assert(offsetof(celeste_prior_t, gal_radius_px_var) == 2744, "celeste_prior_t.gal_radius_px_var");
_Static_assert(sizeof(celeste_group_info_t) == 80, "celeste_group_info_t");
_Static_assert(offsetof(celeste_group_info_t, n_members) == 0, "celeste_group_info_t.n_members");
_Static_assert(offsetof(celeste_group_info_t, n_devices) == 4, "celeste_group_info_t.n_devices");
_Static_assert(offsetof(celeste_group_info_t, exchange) == 8, "celeste_group_info_t.exchange");
_Static_assert(offsetof(celeste_group_info_t, rccl_ranks) == 12, "celeste_group_info_t.rccl_ranks");
_Static_assert(offsetof(celeste_group_info_t, devices) == 16, "celeste_group_info_t.devices");
_Static_assert(sizeof(celeste_problem_t) == 88, "celeste_problem_t");
_Static_assert(offsetof(celeste_problem_t, n_images) == 0, "celeste_problem_t.n_images");
_Static_assert(offsetof(celeste_problem_t, n_sources) == 4, "celeste_problem_t.n_sources");
_Static_assert(offsetof(celeste_problem_t, psf_K) == 8, "celeste_problem_t.psf_K");
_Static_assert(offsetof(celeste_problem_t, n_stamps) == 12, "celeste_problem_t.n_stamps");
_Static_assert(offsetof(celeste_problem_t, images) == 16, "celeste_problem_t.images");
_Static_assert(offsetof(celeste_problem_t, patches) == 24, "celeste_problem_t.patches");
_Static_assert(offsetof(celeste_problem_t, stamps) == 32, "celeste_problem_t.stamps");
_Static_assert(offsetof(celeste_problem_t, nbr_offsets) == 40, "celeste_problem_t.nbr_offsets");
_Static_assert(offsetof(celeste_problem_t, nbr_index) == 48, "celeste_problem_t.nbr_index");
_Static_assert(offsetof(celeste_problem_t, prior) == 56, "celeste_problem_t.prior");
_Static_assert(offsetof(celeste_problem_t, n_patch_entries) == 64, "celeste_problem_t.n_patch_entries");
_Static_assert(offsetof(celeste_problem_t, patch_source) == 72, "celeste_problem_t.patch_source");
_Static_assert(offsetof(celeste_problem_t, patch_image) == 80, "celeste_problem_t.patch_image");
_Static_assert(sizeof(celeste_work_stats_t) == 56, "celeste_work_stats_t");
_Static_assert(offsetof(celeste_work_stats_t, n_targets) == 0, "celeste_work_stats_t.n_targets");
_Static_assert(offsetof(celeste_work_stats_t, active_pixel_visits) == 8, "celeste_work_stats_t.active_pixel_visits");
_Static_assert(offsetof(celeste_work_stats_t, patch_rows) == 16, "celeste_work_stats_t.patch_rows");
_Static_assert(offsetof(celeste_work_stats_t, neighbor_links) == 24, "celeste_work_stats_t.neighbor_links");
_Static_assert(offsetof(celeste_work_stats_t, algorithmic_bytes) == 32, "celeste_work_stats_t.algorithmic_bytes");
_Static_assert(offsetof(celeste_work_stats_t, record_bytes) == 40, "celeste_work_stats_t.record_bytes");
_Static_assert(offsetof(celeste_work_stats_t, record_tiles) == 48, "celeste_work_stats_t.record_tiles");
_Static_assert(sizeof(celeste_optim_config_t) == 72, "celeste_optim_config_t");
_Static_assert(offsetof(celeste_optim_config_t, loc_width) == 0, "celeste_optim_config_t.loc_width");
_Static_assert(offsetof(celeste_optim_config_t, loc_scale) == 8, "celeste_optim_config_t.loc_scale");
_Static_assert(offsetof(celeste_optim_config_t, max_iters) == 16, "celeste_optim_config_t.max_iters");
_Static_assert(offsetof(celeste_optim_config_t, include_kl) == 20, "celeste_optim_config_t.include_kl");
_Static_assert(offsetof(celeste_optim_config_t, xtol_abs) == 24, "celeste_optim_config_t.xtol_abs");
_Static_assert(offsetof(celeste_optim_config_t, ftol_rel) == 32, "celeste_optim_config_t.ftol_rel");
_Static_assert(offsetof(celeste_optim_config_t, gtol) == 40, "celeste_optim_config_t.gtol");
_Static_assert(offsetof(celeste_optim_config_t, initial_delta) == 48, "celeste_optim_config_t.initial_delta");
_Static_assert(offsetof(celeste_optim_config_t, delta_hat) == 56, "celeste_optim_config_t.delta_hat");
_Static_assert(offsetof(celeste_optim_config_t, tr_secular_iters) == 64, "celeste_optim_config_t.tr_secular_iters");

#define P CELESTE_P

/* ---- the fixture: manifest lines `name dtype ndims dim1 [dim2 ...] byte_offset` over one binary blob ---- */
typedef struct { char name[48]; char dtype[16]; int nd; long dims[4]; long off; } entry_t;
static entry_t g_e[128];
static int g_n = 0;
static unsigned char *g_blob = NULL;

static void die(const char *msg) { fprintf(stderr, "cabi_caller: %s\n", msg); exit(2); }

static void load_fixture(const char *stem) {
    char path[1024];
    FILE *f;
    long size;
    snprintf(path, sizeof path, "%s.txt", stem);
    f = fopen(path, "r");
    if (!f) die("cannot open the manifest");
    while (g_n < 128) {
        entry_t *e = &g_e[g_n];
        int k;
        if (fscanf(f, "%47s %15s %d", e->name, e->dtype, &e->nd) != 3) break;
        if (e->nd < 1 || e->nd > 4) die("bad manifest line");
        for (k = 0; k < e->nd; ++k) if (fscanf(f, "%ld", &e->dims[k]) != 1) die("bad manifest line");
        if (fscanf(f, "%ld", &e->off) != 1) die("bad manifest line");
        ++g_n;
    }
    fclose(f);
    snprintf(path, sizeof path, "%s.bin", stem);
    f = fopen(path, "rb");
    if (!f) die("cannot open the blob");
    fseek(f, 0, SEEK_END);
    size = ftell(f);
    fseek(f, 0, SEEK_SET);
    g_blob = (unsigned char *)malloc((size_t)size + 8);
    if (!g_blob || fread(g_blob, 1, (size_t)size, f) != (size_t)size) die("cannot read the blob");
    fclose(f);
}

static const entry_t *find(const char *name, const char *dtype) {
    int k;
    for (k = 0; k < g_n; ++k)
        if (strcmp(g_e[k].name, name) == 0) {
            if (strcmp(g_e[k].dtype, dtype) != 0) die("unexpected dtype in the manifest");
            return &g_e[k];
        }
    fprintf(stderr, "cabi_caller: array %s missing (re-run tests/golden/export_raw.py)\n", name);
    exit(2);
}
static const void *arr(const char *name, const char *dtype) { return g_blob + find(name, dtype)->off; }
static const void *arr_n(const char *fmt, int n, const char *dtype) {
    char name[48];
    snprintf(name, sizeof name, fmt, n);
    return arr(name, dtype);
}

static void check(int st, const char *what) {
    if (st != CELESTE_OK) {
        fprintf(stderr, "cabi_caller: %s -> status %d (%s)\n", what, st, celeste_strerror(st));
        exit(st == CELESTE_ERR_NO_DEVICE ? 5 : 3);
    }
}

static void print_row(const char *tag, int s, const double *x, int n) {
    int k;
    printf("%s %d", tag, s);
    for (k = 0; k < n; ++k) printf(" %.17g", x[k]);
    printf("\n");
}

int main(int argc, char **argv) {
    int N, S, n, s, k, device = 0, members = 1, distinct = 0;
    celeste_image_t *images;
    celeste_patch_t *patches;
    double *stamps;
    const int32_t *box;
    const double *center;
    const double *vp_in;
    celeste_problem_t prob;
    celeste_ctx_t *ctx = NULL;

    if (argc < 2) die("usage: cabi_caller <fixture stem, e.g. tests/golden/raw/sample_two_body> [device [group members [distinct]]]");
    if (argc > 2) device = atoi(argv[2]);
    if (argc > 3) members = atoi(argv[3]);
    if (argc > 4) distinct = atoi(argv[4]);      /* 1: member q on device + q (RCCL between real devices); 0: all on `device` */
    if (members < 1 || members > 16) die("group members: 1 .. 16");
    if (celeste_version() / 100 != CELESTE_ABI_VERSION / 100) die("library / header ABI version mismatch");
    load_fixture(argv[1]);
    N = (int)*(const int64_t *)arr("n_images", "int64");
    S = (int)*(const int64_t *)arr("n_sources", "int64");

    /* Model.Image (image_model.jl:6-38): planes column-major, h fastest, exactly as the fixture stores them */
    images = (celeste_image_t *)calloc((size_t)N, sizeof *images);
    stamps = (double *)malloc((size_t)N * CELESTE_STAMP * CELESTE_STAMP * sizeof(double));
    for (n = 0; n < N; ++n) {
        char name[48];
        const entry_t *e;
        snprintf(name, sizeof name, "pixels_%d", n + 1);
        e = find(name, "float32");
        images[n].H = (int32_t)e->dims[0];
        images[n].W = (int32_t)e->dims[1];
        images[n].band = (int32_t)*(const int64_t *)arr_n("band_%d", n + 1, "int64");
        images[n].pixels = (const float *)arr_n("pixels_%d", n + 1, "float32");
        images[n].sky = (const float *)arr_n("sky_%d", n + 1, "float32");
        images[n].nelec_per_nmgy = (const float *)arr_n("nelec_per_nmgy_%d", n + 1, "float32");
        /* one psfmap stamp per image (ConstantPSFMap): the 51 x 51 matrix, column-major */
        memcpy(stamps + (size_t)n * CELESTE_STAMP * CELESTE_STAMP, arr_n("psf_stamp_%d", n + 1, "float64"),
               CELESTE_STAMP * CELESTE_STAMP * sizeof(double));
    }
    /* Model.ImagePatch (imaged_sources.jl:60-71), dense table [s * n_images + n]; identity WCS in these fixtures */
    box = (const int32_t *)arr("patch_box", "int32");            /* (S N) x 4 column-major: off_h, off_w, H2, W2 */
    center = (const double *)arr("patch_center", "float64");     /* (S N) x 2 column-major */
    patches = (celeste_patch_t *)calloc((size_t)S * N, sizeof *patches);
    for (s = 0; s < S; ++s)
        for (n = 0; n < N; ++n) {
            const int r = s * N + n, SN = S * N;
            celeste_patch_t *p = &patches[r];
            /* psf_<n>: K x 6 column-major in the fixture -> K rows of {alphaBar, xiBar1, xiBar2, tauBar11, tauBar12, tauBar22} */
            const double *pm = (const double *)arr_n("psf_%d", n + 1, "float64");
            const long K = find("psf_1", "float64")->dims[0];
            double *rows = (double *)malloc((size_t)K * 6 * sizeof(double));
            long kk, j;
            for (kk = 0; kk < K; ++kk) for (j = 0; j < 6; ++j) rows[kk * 6 + j] = pm[kk + K * j];
            p->off_h = box[r]; p->off_w = box[r + SN]; p->H2 = box[r + 2 * SN]; p->W2 = box[r + 3 * SN];
            p->bitmap = NULL;                                     /* = !isnan(pixel) */
            p->wcs_jacobian[0] = 1.0; p->wcs_jacobian[1] = 0.0; p->wcs_jacobian[2] = 0.0; p->wcs_jacobian[3] = 1.0;
            p->world_center[0] = p->pixel_center[0] = center[r];
            p->world_center[1] = p->pixel_center[1] = center[r + SN];
            p->psf = rows;
            p->stamp = n;
        }
    memset(&prob, 0, sizeof prob);
    prob.n_images = N; prob.n_sources = S; prob.psf_K = (int32_t)find("psf_1", "float64")->dims[0]; prob.n_stamps = N;
    prob.images = images; prob.patches = patches; prob.stamps = stamps;
    prob.nbr_offsets = (const int64_t *)arr("nbr_offsets", "int64");
    prob.nbr_index = (const int32_t *)arr("nbr_index", "int32");
    prob.prior = NULL;                                            /* the built-in cfg/{star,gal}_prior tables */
    check(celeste_ctx_create(&prob, device, &ctx), "celeste_ctx_create");

    /* vp: S x 44 column-major in the fixture -> row s = source s */
    {
        const double *m = (const double *)arr("vp", "float64");
        double *t = (double *)malloc((size_t)S * P * sizeof(double));
        for (s = 0; s < S; ++s) for (k = 0; k < P; ++k) t[s * P + k] = m[s + S * k];
        vp_in = t;
    }

    {   /* elbo(ea, vp) per source: ElboMaximize.jl:166 */
        const uint32_t flags = CELESTE_FLAG_GRAD | CELESTE_FLAG_HESS | CELESTE_FLAG_KL;
        double *v = (double *)malloc((size_t)S * sizeof(double)), *d = (double *)malloc((size_t)S * P * sizeof(double));
        double *h = (double *)malloc((size_t)S * P * P * sizeof(double));
        double *bv = (double *)malloc((size_t)S * sizeof(double)), *bd = (double *)malloc((size_t)S * P * sizeof(double));
        double *bh = (double *)malloc((size_t)S * P * P * sizeof(double));
        int64_t *cnt = (int64_t *)malloc((size_t)S * 2 * sizeof(int64_t));
        int32_t *tg = (int32_t *)malloc((size_t)S * sizeof(int32_t)), *st = (int32_t *)malloc((size_t)S * sizeof(int32_t));
        for (s = 0; s < S; ++s) {
            int64_t na = -1, ni = -1;
            check(celeste_elbo_eval(ctx, vp_in, s, flags, &v[s], d + (size_t)s * P, h + (size_t)s * P * P, &na, &ni), "celeste_elbo_eval");
            printf("elbo %d %.17g %lld %lld\n", s, v[s], (long long)na, (long long)ni);
            print_row("d", s, d + (size_t)s * P, P);
            print_row("h", s, h + (size_t)s * P * P, P * P);
            tg[s] = s;
        }
        check(celeste_elbo_eval_batch(ctx, vp_in, S, tg, flags, bv, bd, bh, cnt, st), "celeste_elbo_eval_batch");
        printf("batch_equal %d\n", memcmp(v, bv, (size_t)S * sizeof(double)) == 0 && memcmp(d, bd, (size_t)S * P * sizeof(double)) == 0 &&
                                   memcmp(h, bh, (size_t)S * P * P * sizeof(double)) == 0);
    }
    {   /* maximize!(ea, vp, cfg) for every source, neighbours frozen (ElboMaximize.jl:228-242); then the joint-inference
           loop with every source a layer of its own, two sweeps (ParallelRun.jl:302-397) */
        celeste_optim_config_t cfg;
        double *vp = (double *)malloc((size_t)S * P * sizeof(double)), *el = (double *)malloc((size_t)2 * S * sizeof(double));
        int32_t *tg = (int32_t *)malloc((size_t)2 * S * sizeof(int32_t)), *it = (int32_t *)malloc((size_t)2 * S * sizeof(int32_t));
        int32_t *ev = (int32_t *)malloc((size_t)2 * S * sizeof(int32_t)), *st = (int32_t *)malloc((size_t)2 * S * sizeof(int32_t));
        int64_t *off = (int64_t *)malloc((size_t)(2 * S + 1) * sizeof(int64_t));
        memset(&cfg, 0, sizeof cfg);
        cfg.loc_width = 1e-4; cfg.loc_scale = 1.0; cfg.max_iters = 4; cfg.include_kl = 1; cfg.xtol_abs = 1e-7;
        cfg.ftol_rel = 1e-6; cfg.gtol = 1e-8; cfg.initial_delta = 1.0; cfg.delta_hat = 1e9; cfg.tr_secular_iters = 0;
        memcpy(vp, vp_in, (size_t)S * P * sizeof(double));
        for (s = 0; s < S; ++s) tg[s] = s;
        check(celeste_maximize_batch(ctx, vp, NULL, NULL, S, tg, &cfg, it, ev, el, st), "celeste_maximize_batch");
        for (s = 0; s < S; ++s) {
            printf("maximize %d %d %d %d %.17g", s, (int)it[s], (int)ev[s], (int)st[s], el[s]);
            for (k = 0; k < P; ++k) printf(" %.17g", vp[s * P + k]);
            printf("\n");
        }
        memcpy(vp, vp_in, (size_t)S * P * sizeof(double));
        for (k = 0; k < 2 * S; ++k) { tg[k] = k % S; off[k] = k; }
        off[2 * S] = 2 * S;
        check(celeste_joint_infer(ctx, vp, 2 * S, off, tg, NULL, &cfg, it, ev, el, st), "celeste_joint_infer");
        for (k = 0; k < 2 * S; ++k) printf("joint %d %d %d %d %d %.17g\n", k, (int)tg[k], (int)it[k], (int)ev[k], (int)st[k], el[k]);
        for (s = 0; s < S; ++s) print_row("joint_vp", s, vp + (size_t)s * P, P);
    }
    {   /* the same three computations through a device group: `members` members, all on `device` (one member: RCCL with one
           rank; several on one device: device-to-device copies -- see the header) */
        const uint32_t flags = CELESTE_FLAG_GRAD | CELESTE_FLAG_HESS | CELESTE_FLAG_KL;
        celeste_group_t *grp = NULL;
        celeste_group_info_t gi;
        celeste_optim_config_t cfg;
        int32_t devs[16];
        const size_t nv = (size_t)S, nd = (size_t)S * P, nh = (size_t)S * P * P;
        double *v[2], *d[2], *h[2], *vp[2], *el[2];
        int64_t *cnt[2], *off = (int64_t *)malloc((size_t)(2 * S + 1) * sizeof(int64_t)), n_exch = -1;
        int32_t *st[2], *it[2], *ev[2], *tg = (int32_t *)malloc((size_t)2 * S * sizeof(int32_t));
        int q, eq_eval, eq_max, eq_joint;
        for (q = 0; q < members; ++q) devs[q] = distinct ? device + q : device;
        for (q = 0; q < 2; ++q) {
            v[q] = (double *)calloc(nv, sizeof(double)); d[q] = (double *)calloc(nd, sizeof(double)); h[q] = (double *)calloc(nh, sizeof(double));
            vp[q] = (double *)malloc(nd * sizeof(double)); el[q] = (double *)calloc(2 * nv, sizeof(double));
            cnt[q] = (int64_t *)calloc(2 * nv, sizeof(int64_t)); st[q] = (int32_t *)calloc(2 * nv, sizeof(int32_t));
            it[q] = (int32_t *)calloc(2 * nv, sizeof(int32_t)); ev[q] = (int32_t *)calloc(2 * nv, sizeof(int32_t));
        }
        memset(&cfg, 0, sizeof cfg);
        cfg.loc_width = 1e-4; cfg.loc_scale = 1.0; cfg.max_iters = 4; cfg.include_kl = 1; cfg.xtol_abs = 1e-7;
        cfg.ftol_rel = 1e-6; cfg.gtol = 1e-8; cfg.initial_delta = 1.0; cfg.delta_hat = 1e9; cfg.tr_secular_iters = 0;
        check(celeste_group_create(&prob, members, devs, &grp), "celeste_group_create");
        check(celeste_group_info(grp, &gi), "celeste_group_info");
        printf("group %d %d %d\n", (int)gi.n_members, (int)gi.exchange, (int)gi.rccl_ranks);
        for (s = 0; s < S; ++s) tg[s] = s;
        check(celeste_elbo_eval_batch(ctx, vp_in, S, tg, flags, v[0], d[0], h[0], cnt[0], st[0]), "celeste_elbo_eval_batch");
        check(celeste_group_elbo_eval_batch(grp, vp_in, S, tg, flags, v[1], d[1], h[1], cnt[1], st[1]), "celeste_group_elbo_eval_batch");
        eq_eval = memcmp(v[0], v[1], nv * sizeof(double)) == 0 && memcmp(d[0], d[1], nd * sizeof(double)) == 0 &&
                  memcmp(h[0], h[1], nh * sizeof(double)) == 0 && memcmp(cnt[0], cnt[1], 2 * nv * sizeof(int64_t)) == 0 &&
                  memcmp(st[0], st[1], nv * sizeof(int32_t)) == 0;
        memcpy(vp[0], vp_in, nd * sizeof(double)); memcpy(vp[1], vp_in, nd * sizeof(double));
        check(celeste_maximize_batch(ctx, vp[0], NULL, NULL, S, tg, &cfg, it[0], ev[0], el[0], st[0]), "celeste_maximize_batch");
        check(celeste_group_maximize_batch(grp, vp[1], NULL, NULL, S, tg, &cfg, it[1], ev[1], el[1], st[1]), "celeste_group_maximize_batch");
        eq_max = memcmp(vp[0], vp[1], nd * sizeof(double)) == 0 && memcmp(it[0], it[1], nv * sizeof(int32_t)) == 0 &&
                 memcmp(ev[0], ev[1], nv * sizeof(int32_t)) == 0 && memcmp(el[0], el[1], nv * sizeof(double)) == 0 &&
                 memcmp(st[0], st[1], nv * sizeof(int32_t)) == 0;
        /* joint inference, two sweeps: every source a batch of its own with one one-source component (the sources of these
           fixtures are neighbours) == every source a layer of its own, twice */
        memcpy(vp[0], vp_in, nd * sizeof(double)); memcpy(vp[1], vp_in, nd * sizeof(double));
        for (k = 0; k < 2 * S; ++k) { tg[k] = k % S; off[k] = k; }
        off[2 * S] = 2 * S;
        check(celeste_joint_infer(ctx, vp[0], 2 * S, off, tg, NULL, &cfg, it[0], ev[0], el[0], st[0]), "celeste_joint_infer");
        check(celeste_group_joint_infer(grp, vp[1], 2, S, off, off, tg, NULL, &cfg, it[1], ev[1], el[1], st[1], &n_exch),
              "celeste_group_joint_infer");
        eq_joint = memcmp(vp[0], vp[1], nd * sizeof(double)) == 0 && memcmp(it[0], it[1], 2 * nv * sizeof(int32_t)) == 0 &&
                   memcmp(ev[0], ev[1], 2 * nv * sizeof(int32_t)) == 0 && memcmp(el[0], el[1], 2 * nv * sizeof(double)) == 0 &&
                   memcmp(st[0], st[1], 2 * nv * sizeof(int32_t)) == 0;
        printf("group_equal %d %d %d %lld\n", eq_eval, eq_max, eq_joint, (long long)n_exch);
        celeste_group_destroy(grp);
    }
    celeste_ctx_destroy(ctx);
    return 0;
}
