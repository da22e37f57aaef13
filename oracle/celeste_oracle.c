/*
 * celeste_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, single-threaded-per-call, fp64 restatement of the reference's ELBO
 * hot path (Celeste.jl @ /root/reference, Julia 0.6).  It follows the reference
 * function by function -- dense 44-vector / 44x44 SensitiveFloats per pixel,
 * 28 bivariate-normal components per galaxy evaluation, per-component
 * (x, Sigma) -> (pos, shape) derivative transforms -- so that it can serve as
 *   (1) the parity checker for the HIP path (tests/, __graft_entry__.smoke()),
 *   (2) the "port" CPU baseline timed by bench.py (cpu_baseline leg only).
 * Nothing under celeste.jl_amd/ may include, link or call this file.
 *
 * PARITY STATUS: the reference cannot be compiled or imported here (Julia is
 * absent) and its tests hold no numeric ELBO goldens, so the
 * whole-ELBO value is "parity unpinned".  What IS pinned (tests/test_oracle_*.py):
 * the closed-form known answers of the reference's own tests (test_elbo.jl:45-61,
 * test_psf.jl:121-135, test_kl.jl:30-72, Appendix C of SURVEY.md), and the
 * reference's central property "hand-written derivatives == automatic
 * differentiation of the same value function" (test_elbo.jl:223-301), here
 * against torch.autograd on an independently written value-only restatement.
 * Third-party arithmetic restated from its published algorithm: Interpolations.jl
 * BSpline(Cubic(Line())), OnGrid() (REQUIRE:21, unpinned version) -- natural
 * bicubic spline, cell index clamped to [1,50].
 *
 * Each function cites the reference file:line it restates.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/celeste_mi355x.h"
#include "patch_lookup.h"

#define P 44
#define NB 5

static const celeste_prior_t ORACLE_DEFAULT_PRIOR =
#include "../celeste.jl_amd/csrc/prior_tables.inc"
    ;

/* ---- parameter indices, 0-based (src/model/param_set.jl:76-107) ---------- */
enum {
    ID_POS = 0, ID_DEV = 2, ID_RATIO = 3, ID_ANGLE = 4, ID_RADIUS = 5,
    ID_FLUX_LOC = 6, ID_FLUX_SCALE = 8, ID_COLOR_MEAN = 10, ID_COLOR_VAR = 18,
    ID_IS_STAR = 26, ID_K = 28
};
/* bids (param_set.jl:63-74), 0-based: flux_loc 0, flux_scale 1, color_mean 2..5, color_var 6..9 */

/* brightness_standard_alignment[i] (param_set.jl:163-164) */
static void bright_ids(int i, int out[10]) {
    out[0] = ID_FLUX_LOC + i;
    out[1] = ID_FLUX_SCALE + i;
    for (int c = 0; c < 4; ++c) out[2 + c] = ID_COLOR_MEAN + 4 * i + c;
    for (int c = 0; c < 4; ++c) out[6 + c] = ID_COLOR_VAR + 4 * i + c;
}

/* ---- galaxy prototypes (src/model/light_source_model.jl:45-75) ---------- */
static double G_ETA[2][8], G_NU[2][8];
static int g_proto_ready = 0;
static void init_prototypes(void) {
    if (g_proto_ready) return;
    const double dev_amp[8] = {4.26347652e-2, 2.40127183e-1, 6.85907632e-1, 1.51937350,
                               2.83627243, 4.46467501, 5.72440830, 5.60989349};
    const double dev_var[8] = {2.23759216e-4, 1.00220099e-3, 4.18731126e-3, 1.69432589e-2,
                               6.84850479e-2, 2.87207080e-1, 1.33320254, 8.40215071};
    const double exp_amp[6] = {2.34853813e-3, 3.07995260e-2, 2.23364214e-1,
                               1.17949102, 4.33873750, 5.99820770};
    const double exp_var[6] = {1.20078965e-3, 8.84526493e-3, 3.91463084e-2,
                               1.39976817e-1, 4.60962500e-1, 1.50159566};
    const double er[2] = {1.078031, 0.928896};
    double sd = 0, se = 0;
    for (int j = 0; j < 8; ++j) sd += dev_amp[j];
    for (int j = 0; j < 6; ++j) se += exp_amp[j];
    for (int j = 0; j < 8; ++j) { G_ETA[0][j] = dev_amp[j] / sd; G_NU[0][j] = dev_var[j] / (er[0] * er[0]); }
    for (int j = 0; j < 8; ++j) { G_ETA[1][j] = 0; G_NU[1][j] = 0; }
    for (int j = 0; j < 6; ++j) { G_ETA[1][j] = exp_amp[j] / se; G_NU[1][j] = exp_var[j] / (er[1] * er[1]); }
    g_proto_ready = 1;
}
void celeste_oracle_galaxy_prototypes(double eta[16], double nu[16]) {
    init_prototypes();
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 8; ++j) { eta[8 * i + j] = G_ETA[i][j]; nu[8 * i + j] = G_NU[i][j]; }
}

/* ---- SensitiveFloat (src/SensitiveFloats.jl:23-47) ------------------------ */
typedef struct { int p; double v; double *d; double *h; } SF; /* h col-major p x p */

static SF sf_new(int p) {
    SF s; s.p = p; s.v = 0;
    s.d = (double *)calloc((size_t)p, sizeof(double));
    s.h = (double *)calloc((size_t)p * p, sizeof(double));
    return s;
}
static void sf_free(SF *s) { free(s->d); free(s->h); }
static void sf_zero(SF *s) { /* zero! SensitiveFloats.jl:83-93 */
    s->v = 0; memset(s->d, 0, sizeof(double) * s->p); memset(s->h, 0, sizeof(double) * s->p * s->p);
}
#define H_(s, i, j) ((s)->h[(i) + (size_t)(s)->p * (j)])

/* combine_sfs! + combine_sfs_hessian! (SensitiveFloats.jl:99-165): r may alias sf1 */
static void combine_sfs(const SF *sf1, const SF *sf2, SF *r, double v, const double g_d[2],
                        const double g_h[4], int has_grad, int has_hess) {
    const int p = r->p;
    if (has_hess) {
        for (int i2 = 0; i2 < p; ++i2) {
            double f11 = g_h[0] * sf1->d[i2] + g_h[2] * sf2->d[i2];
            double f21 = g_h[2] * sf1->d[i2] + g_h[3] * sf2->d[i2];
            for (int i1 = 0; i1 < p; ++i1) {
                H_(r, i1, i2) = g_d[0] * H_(sf1, i1, i2) + g_d[1] * H_(sf2, i1, i2) +
                                f11 * sf1->d[i1] + f21 * sf2->d[i1];
            }
        }
    }
    if (has_grad)
        for (int i = 0; i < p; ++i) r->d[i] = g_d[0] * sf1->d[i] + g_d[1] * sf2->d[i];
    r->v = v;
}
/* multiply_sfs! (SensitiveFloats.jl:174-179) */
static void multiply_sfs(SF *sf1, const SF *sf2) {
    double v = sf1->v * sf2->v;
    double g_d[2] = {sf2->v, sf1->v};
    const double g_h[4] = {0, 1, 1, 0};
    combine_sfs(sf1, sf2, sf1, v, g_d, g_h, 1, 1);
}
static void set_hess(SF *s, int i, int j, double v) { H_(s, i, j) = v; H_(s, j, i) = v; }

/* ---- SourceBrightness (src/deterministic_vi/source_brightness.jl:27-202) -- */
typedef struct { SF E_l_a[NB][2]; SF E_ll_a[NB][2]; } SourceBrightness;

static void sb_load(SourceBrightness *sb, const double *vs, int derivs) {
    for (int i = 0; i < 2; ++i) {
        double flux_loc = vs[ID_FLUX_LOC + i], flux_scale = vs[ID_FLUX_SCALE + i];
        const double *cm = vs + ID_COLOR_MEAN + 4 * i, *cv = vs + ID_COLOR_VAR + 4 * i;
        SF *E = NULL;
        for (int b = 0; b < NB; ++b) { sb->E_l_a[b][i] = sf_new(10); sb->E_ll_a[b][i] = sf_new(10); }
        /* band indices 0-based: reference band 3 -> 2 */
        sb->E_l_a[2][i].v = exp(flux_loc + 0.5 * flux_scale);
        sb->E_l_a[3][i].v = exp(cm[2] + .5 * cv[2]);
        sb->E_l_a[4][i].v = exp(cm[3] + .5 * cv[3]);
        sb->E_l_a[1][i].v = exp(-cm[1] + .5 * cv[1]);
        sb->E_l_a[0][i].v = exp(-cm[0] + .5 * cv[0]);
        if (derivs) {
            E = &sb->E_l_a[2][i];
            E->d[0] = E->v; E->d[1] = E->v * .5;
            set_hess(E, 0, 0, E->v); set_hess(E, 0, 1, E->v * 0.5); set_hess(E, 1, 1, E->v * 0.25);
            /* band 4 = band 3 * color 3 */
            E = &sb->E_l_a[3][i];
            E->d[2 + 2] = E->v; E->d[6 + 2] = E->v * .5;
            set_hess(E, 4, 4, E->v); set_hess(E, 4, 8, E->v * 0.5); set_hess(E, 8, 8, E->v * 0.25);
            multiply_sfs(E, &sb->E_l_a[2][i]);
            /* band 5 = band 4 * color 4 */
            E = &sb->E_l_a[4][i];
            E->d[2 + 3] = E->v; E->d[6 + 3] = E->v * .5;
            set_hess(E, 5, 5, E->v); set_hess(E, 5, 9, E->v * 0.5); set_hess(E, 9, 9, E->v * 0.25);
            multiply_sfs(E, &sb->E_l_a[3][i]);
            /* band 2 = band 3 * color 2 */
            E = &sb->E_l_a[1][i];
            E->d[2 + 1] = E->v * -1.; E->d[6 + 1] = E->v * .5;
            set_hess(E, 3, 3, E->v); set_hess(E, 3, 7, E->v * -0.5); set_hess(E, 7, 7, E->v * 0.25);
            multiply_sfs(E, &sb->E_l_a[2][i]);
            /* band 1 = band 2 * color 1 */
            E = &sb->E_l_a[0][i];
            E->d[2 + 0] = E->v * -1.; E->d[6 + 0] = E->v * .5;
            set_hess(E, 2, 2, E->v); set_hess(E, 2, 6, E->v * -0.5); set_hess(E, 6, 6, E->v * 0.25);
            multiply_sfs(E, &sb->E_l_a[1][i]);
        } else {
            sb->E_l_a[3][i].v *= sb->E_l_a[2][i].v;
            sb->E_l_a[4][i].v *= sb->E_l_a[3][i].v;
            sb->E_l_a[1][i].v *= sb->E_l_a[2][i].v;
            sb->E_l_a[0][i].v *= sb->E_l_a[1][i].v;
        }
        /* squared terms */
        sb->E_ll_a[2][i].v = exp(2 * flux_loc + 2 * flux_scale);
        sb->E_ll_a[3][i].v = exp(2 * cm[2] + 2 * cv[2]);
        sb->E_ll_a[4][i].v = exp(2 * cm[3] + 2 * cv[3]);
        sb->E_ll_a[1][i].v = exp(-2 * cm[1] + 2 * cv[1]);
        sb->E_ll_a[0][i].v = exp(-2 * cm[0] + 2 * cv[0]);
        if (derivs) {
            E = &sb->E_ll_a[2][i];
            E->d[0] = 2 * E->v; E->d[1] = 2 * E->v;
            set_hess(E, 0, 0, 4.0 * E->v); set_hess(E, 0, 1, 4.0 * E->v); set_hess(E, 1, 1, 4.0 * E->v);
            E = &sb->E_ll_a[3][i];
            E->d[4] = E->v * 2.; E->d[8] = E->v * 2.;
            set_hess(E, 4, 4, E->v * 4.0); set_hess(E, 4, 8, E->v * 4.0); set_hess(E, 8, 8, E->v * 4.0);
            multiply_sfs(E, &sb->E_ll_a[2][i]);
            E = &sb->E_ll_a[4][i];
            E->d[5] = E->v * 2.; E->d[9] = E->v * 2.;
            set_hess(E, 5, 5, E->v * 4.0); set_hess(E, 5, 9, E->v * 4.0); set_hess(E, 9, 9, E->v * 4.0);
            multiply_sfs(E, &sb->E_ll_a[3][i]);
            E = &sb->E_ll_a[1][i];
            E->d[3] = E->v * -2.; E->d[7] = E->v * 2.;
            set_hess(E, 3, 3, E->v * 4.0); set_hess(E, 7, 7, E->v * 4.0); set_hess(E, 3, 7, E->v * -4.0);
            multiply_sfs(E, &sb->E_ll_a[2][i]);
            E = &sb->E_ll_a[0][i];
            E->d[2] = E->v * -2.; E->d[6] = E->v * 2.;
            set_hess(E, 2, 2, E->v * 4.0); set_hess(E, 6, 6, E->v * 4.0); set_hess(E, 2, 6, E->v * -4.0);
            multiply_sfs(E, &sb->E_ll_a[1][i]);
        } else {
            sb->E_ll_a[3][i].v *= sb->E_ll_a[2][i].v;
            sb->E_ll_a[4][i].v *= sb->E_ll_a[3][i].v;
            sb->E_ll_a[1][i].v *= sb->E_ll_a[2][i].v;
            sb->E_ll_a[0][i].v *= sb->E_ll_a[1][i].v;
        }
    }
}
static void sb_free(SourceBrightness *sb) {
    for (int b = 0; b < NB; ++b) for (int i = 0; i < 2; ++i) { sf_free(&sb->E_l_a[b][i]); sf_free(&sb->E_ll_a[b][i]); }
}
/* exported for known-answer tests: values only */
void celeste_oracle_source_brightness(const double *vs, double E_l[10], double E_ll[10]) {
    SourceBrightness sb; sb_load(&sb, vs, 1);
    for (int b = 0; b < NB; ++b) for (int i = 0; i < 2; ++i) { E_l[b + NB * i] = sb.E_l_a[b][i].v; E_ll[b + NB * i] = sb.E_ll_a[b][i].v; }
    sb_free(&sb);
}

/* ---- BivariateNormals.jl ---------------------------------------------------- */
/* get_bvn_cov (BivariateNormals.jl:29-43); out col-major 2x2 */
void celeste_oracle_get_bvn_cov(double ab, double angle, double scale, double out[4]) {
    double cp = cos(angle), sp = sin(angle);
    double ab_term = ab * ab - 1;
    double scale_squared = scale * scale;
    double off = -scale_squared * cp * sp * ab_term;
    out[0] = scale_squared * (1 + ab_term * (sp * sp));
    out[1] = off; out[2] = off;
    out[3] = scale_squared * (1 + ab_term * (cp * cp));
}

/* BvnComponent (BivariateNormals.jl:143-191) */
typedef struct {
    double mean[2]; double prec[4]; /* col-major */
    double z; double dsiginv_dsig[9]; /* col-major 3x3 [a + 3 b] */
} Bvn;

static void bvn_make(Bvn *b, const double mean[2], const double cov[4], double weight, int siginv_deriv) {
    double det = cov[0] * cov[3] - cov[2] * cov[1];
    double c = 1 / (sqrt(det) * 2 * M_PI);
    double idet = 1 / det;
    b->mean[0] = mean[0]; b->mean[1] = mean[1];
    b->prec[0] = cov[3] * idet; b->prec[1] = -cov[1] * idet; b->prec[2] = -cov[2] * idet; b->prec[3] = cov[0] * idet;
    b->z = c * weight;
    memset(b->dsiginv_dsig, 0, sizeof b->dsiginv_dsig);
    if (siginv_deriv) {
        double p11 = b->prec[0], p12 = b->prec[2], p21 = b->prec[1], p22 = b->prec[3];
        double d11 = -p11 * p11, d12 = -2 * p11 * p12, d13 = -p12 * p12;
        double d21 = -p11 * p21, d22 = -(p11 * p22 + p12 * p12), d23 = -p22 * p12;
        double d32 = -2 * p22 * p21, d33 = -p22 * p22;
        /* row 3 col 1 is written with d13 in the reference (BivariateNormals.jl:181) */
        double m[3][3] = {{d11, d12, d13}, {d21, d22, d23}, {d13, d32, d33}};
        for (int a = 0; a < 3; ++a) for (int c2 = 0; c2 < 3; ++c2) b->dsiginv_dsig[a + 3 * c2] = m[a][c2];
    }
}

/* BivariateNormalDerivatives (BivariateNormals.jl:51-108) */
typedef struct {
    double py1, py2, f_pre;
    double bvn_x_d[2], bvn_sig_d[3], bvn_xx_h[4], bvn_xsig_h[6] /*2x3*/, bvn_sigsig_h[9];
    double dpy1_dsig[3], dpy2_dsig[3];
    double bvn_u_d[2], bvn_uu_h[4], bvn_s_d[3], bvn_ss_h[9], bvn_us_h[6] /*2x3*/;
} BvnDerivs;

/* eval_bvn_pdf! (BivariateNormals.jl:208-222) */
static void eval_bvn_pdf(BvnDerivs *bd, const Bvn *b, const double x[2]) {
    double dx1 = x[0] - b->mean[0], dx2 = x[1] - b->mean[1];
    bd->py1 = b->prec[0] * dx1 + b->prec[2] * dx2;
    bd->py2 = b->prec[1] * dx1 + b->prec[3] * dx2;
    bd->f_pre = b->z * exp(-0.5 * (dx1 * bd->py1 + dx2 * bd->py2));
}

/* get_bvn_derivs! (BivariateNormals.jl:240-319) */
static void get_bvn_derivs(BvnDerivs *bd, const Bvn *b, int x_hess, int sig_hess) {
    double p11 = b->prec[0], p12 = b->prec[2], p22 = b->prec[3];
    double py1 = bd->py1, py2 = bd->py2;
    bd->bvn_x_d[0] = -py1; bd->bvn_x_d[1] = -py2;
    if (x_hess) {
        bd->bvn_xx_h[0] = -p11; bd->bvn_xx_h[3] = -p22; bd->bvn_xx_h[1] = bd->bvn_xx_h[2] = -p12;
    }
    bd->bvn_sig_d[0] = 0.5 * py1 * py1 - 0.5 * p11;
    bd->bvn_sig_d[1] = py1 * py2 - p12;
    bd->bvn_sig_d[2] = 0.5 * py2 * py2 - 0.5 * p22;
    if (sig_hess) {
        bd->dpy1_dsig[0] = -py1 * p11;
        bd->dpy1_dsig[1] = -py2 * p11 - py1 * p12;
        bd->dpy1_dsig[2] = -py2 * p12;
        bd->dpy2_dsig[0] = -py1 * p12;
        bd->dpy2_dsig[1] = -py1 * p22 - py2 * p12;
        bd->dpy2_dsig[2] = -py2 * p22;
        for (int s = 0; s < 3; ++s) {
            bd->bvn_sigsig_h[0 + 3 * s] = py1 * bd->dpy1_dsig[s] - 0.5 * b->dsiginv_dsig[0 + 3 * s];
            bd->bvn_sigsig_h[1 + 3 * s] = py1 * bd->dpy2_dsig[s] + py2 * bd->dpy1_dsig[s] - b->dsiginv_dsig[1 + 3 * s];
            bd->bvn_sigsig_h[2 + 3 * s] = py2 * bd->dpy2_dsig[s] - 0.5 * b->dsiginv_dsig[2 + 3 * s];
        }
        for (int xi = 0; xi < 2; ++xi) {
            /* precision[1, x_ind], precision[2, x_ind] */
            double pr1 = b->prec[0 + 2 * xi], pr2 = b->prec[1 + 2 * xi];
            bd->bvn_xsig_h[xi + 2 * 0] = py1 * pr1;
            bd->bvn_xsig_h[xi + 2 * 1] = py1 * pr2 + py2 * pr1;
            bd->bvn_xsig_h[xi + 2 * 2] = py2 * pr2;
        }
    }
}

/* GalaxySigmaDerivs (BivariateNormals.jl:331-397).  NB argument order. */
typedef struct { double j[9]; /* 3x3 col-major [sig + 3 shape] */ double t[27]; /* [sig + 3 s1 + 9 s2] */ } SigSF;

static void sigsf_make(SigSF *s, double gal_angle, double gal_axis_ratio, double gal_radius_px,
                       const double XiXi[4], double nuBar, int tensor) {
    double cos_sin = cos(gal_angle) * sin(gal_angle);
    double sin_sq = sin(gal_angle) * sin(gal_angle);
    double cos_sq = cos(gal_angle) * cos(gal_angle);
    double r2 = gal_radius_px * gal_radius_px, ab = gal_axis_ratio;
    double j[9];
    double c1 = 2 * ab * r2;
    j[0] = c1 * sin_sq; j[1] = c1 * -cos_sin; j[2] = c1 * cos_sq;
    double c2 = r2 * (ab * ab - 1);
    j[3] = c2 * (2 * cos_sin); j[4] = c2 * (sin_sq - cos_sq); j[5] = c2 * (-2 * cos_sin);
    /* XiXi[1], XiXi[2], XiXi[4] (linear indices) */
    j[6] = 2 * XiXi[0] / gal_radius_px; j[7] = 2 * XiXi[1] / gal_radius_px; j[8] = 2 * XiXi[3] / gal_radius_px;
    double t[27];
    memset(t, 0, sizeof t);
    if (tensor) {
        double a = 2 * r2;
        t[0] = sin_sq * a; t[1] = -cos_sin * a; t[2] = cos_sq * a;                       /* t[:,1,1] */
        t[3] = 2 * cos_sin * a * ab; t[4] = (sin_sq - cos_sq) * a * ab; t[5] = -2 * cos_sin * a * ab; /* t[:,2,1] */
        t[6] = 2 * j[0] / gal_radius_px; t[7] = 2 * j[1] / gal_radius_px; t[8] = 2 * j[2] / gal_radius_px; /* t[:,3,1] */
        t[9] = t[3]; t[10] = t[4]; t[11] = t[5];                                          /* t[:,1,2] */
        double b2 = a * (ab * ab - 1);
        t[12] = (cos_sq - sin_sq) * b2; t[13] = 2 * cos_sin * b2; t[14] = (sin_sq - cos_sq) * b2; /* t[:,2,2] */
        t[15] = 2 * j[3] / gal_radius_px; t[16] = 2 * j[4] / gal_radius_px; t[17] = 2 * j[5] / gal_radius_px; /* t[:,3,2] */
        t[18] = t[6]; t[19] = t[7]; t[20] = t[8];                                         /* t[:,1,3] */
        t[21] = t[15]; t[22] = t[16]; t[23] = t[17];                                      /* t[:,2,3] */
        /* XiXi[1 << (k-1)] = XiXi[1], XiXi[2], XiXi[4] */
        t[24] = 2 * XiXi[0] / r2; t[25] = 2 * XiXi[1] / r2; t[26] = 2 * XiXi[3] / r2;     /* t[:,3,3] */
    }
    for (int k = 0; k < 9; ++k) s->j[k] = j[k] * nuBar;
    for (int k = 0; k < 27; ++k) s->t[k] = t[k] * nuBar;
}

/* transform_bvn_ux_derivs! (BivariateNormals.jl:414-448); J col-major 2x2 */
static void transform_bvn_ux_derivs(BvnDerivs *bd, const double J[4], int hess) {
    bd->bvn_u_d[0] = -(bd->bvn_x_d[0] * J[0] + bd->bvn_x_d[1] * J[1]);
    bd->bvn_u_d[1] = -(bd->bvn_x_d[0] * J[2] + bd->bvn_x_d[1] * J[3]);
    if (hess) {
        memset(bd->bvn_uu_h, 0, sizeof bd->bvn_uu_h);
        for (int x2 = 0; x2 < 2; ++x2) for (int x1 = 0; x1 < 2; ++x1) for (int u2 = 0; u2 < 2; ++u2) {
            double inner = bd->bvn_xx_h[x1 + 2 * x2] * J[x2 + 2 * u2];
            for (int u1 = 0; u1 <= u2; ++u1) bd->bvn_uu_h[u1 + 2 * u2] += inner * J[x1 + 2 * u1];
        }
        bd->bvn_uu_h[1 + 2 * 0] = bd->bvn_uu_h[0 + 2 * 1];
    }
}

/* transform_bvn_derivs! + transform_bvn_derivs_hessian! (BivariateNormals.jl:465-572) */
static void transform_bvn_derivs(BvnDerivs *bd, const SigSF *s, const double J[4], int hess) {
    transform_bvn_ux_derivs(bd, J, hess);
    for (int sh = 0; sh < 3; ++sh) {
        bd->bvn_s_d[sh] = 0;
        for (int sg = 0; sg < 3; ++sg) bd->bvn_s_d[sh] += bd->bvn_sig_d[sg] * s->j[sg + 3 * sh];
    }
    if (!hess) return;
    memset(bd->bvn_ss_h, 0, sizeof bd->bvn_ss_h);
    memset(bd->bvn_us_h, 0, sizeof bd->bvn_us_h);
    for (int s2 = 0; s2 < 3; ++s2) for (int s1 = 0; s1 <= s2; ++s1) for (int sg = 0; sg < 3; ++sg)
        bd->bvn_ss_h[s1 + 3 * s2] += bd->bvn_sig_d[sg] * s->t[sg + 3 * s1 + 9 * s2];
    for (int g1 = 0; g1 < 3; ++g1) for (int g2 = 0; g2 < 3; ++g2) for (int s2 = 0; s2 < 3; ++s2) {
        double inner = bd->bvn_sigsig_h[g1 + 3 * g2] * s->j[g2 + 3 * s2];
        for (int s1 = 0; s1 <= s2; ++s1) bd->bvn_ss_h[s1 + 3 * s2] += inner * s->j[g1 + 3 * s1];
    }
    for (int s2 = 0; s2 < 3; ++s2) for (int s1 = 0; s1 <= s2; ++s1) bd->bvn_ss_h[s2 + 3 * s1] = bd->bvn_ss_h[s1 + 3 * s2];
    for (int sh = 0; sh < 3; ++sh) for (int u = 0; u < 2; ++u) for (int sg = 0; sg < 3; ++sg) for (int x = 0; x < 2; ++x)
        bd->bvn_us_h[u + 2 * sh] += bd->bvn_xsig_h[x + 2 * sg] * s->j[sg + 3 * sh] * (-J[x + 2 * u]);
}

/* ---- GalaxyCacheComponent / load_bvn_mixtures! (src/model/fsm_util.jl:29-169) */
typedef struct { double dir, frac_i; Bvn bmc; SigSF sig; int used; } GalComp;

static void linear_world_to_pix(const double J[4], const double wc[2], const double pc[2],
                                const double world[2], double out[2]) { /* wcs_utils.jl:14-18 */
    double d0 = world[0] - wc[0], d1 = world[1] - wc[1];
    out[0] = J[0] * d0 + J[2] * d1 + pc[0];
    out[1] = J[1] * d0 + J[3] * d1 + pc[1];
}

/* gal_mcs[k + K*(j + 8*i)] for one source in one image */
static void load_bvn_mixtures_source(GalComp *mcs, const celeste_patch_t *p, int psf_K,
                                     const double *sp, int calc_grad, int calc_hess) {
    init_prototypes();
    double m_pos[2];
    linear_world_to_pix(p->wcs_jacobian, p->world_center, p->pixel_center, sp + ID_POS, m_pos);
    for (int i = 0; i < 2; ++i) {
        double dir = (i == 0) ? 1. : -1.;
        double frac_i = (i == 0) ? sp[ID_DEV] : 1. - sp[ID_DEV];
        int nj = (i == 0) ? 8 : 6;
        for (int j = 0; j < 8; ++j) for (int k = 0; k < psf_K; ++k) mcs[k + psf_K * (j + 8 * i)].used = 0;
        for (int j = 0; j < nj; ++j) for (int k = 0; k < psf_K; ++k) {
            GalComp *g = &mcs[k + psf_K * (j + 8 * i)];
            const double *pc = p->psf + 6 * k;
            double XiXi[4];
            celeste_oracle_get_bvn_cov(sp[ID_RATIO], sp[ID_ANGLE], sp[ID_RADIUS], XiXi);
            double mean_s[2] = {pc[1] + m_pos[0], pc[2] + m_pos[1]};
            double nu = G_NU[i][j];
            double var_s[4] = {pc[3] + nu * XiXi[0], pc[4] + nu * XiXi[1], pc[4] + nu * XiXi[2], pc[5] + nu * XiXi[3]};
            double weight = pc[0] * G_ETA[i][j];
            bvn_make(&g->bmc, mean_s, var_s, weight, calc_grad && calc_hess);
            if (calc_grad) sigsf_make(&g->sig, sp[ID_ANGLE], sp[ID_RATIO], sp[ID_RADIUS], XiXi, nu, calc_hess);
            else memset(&g->sig, 0, sizeof g->sig);
            g->dir = dir; g->frac_i = frac_i; g->used = 1;
        }
    }
}

/* accum_galaxy_pos! (fsm_util.jl:255-346); fs1m has p = 6 */
static void accum_galaxy_pos(SF *fs1m, BvnDerivs *bd, const GalComp *g, const double x[2], const double J[4],
                             int active, int has_grad, int has_hess) {
    eval_bvn_pdf(bd, &g->bmc, x);
    double f = bd->f_pre * g->frac_i;
    fs1m->v += f;
    if (!(has_grad && active)) return;
    get_bvn_derivs(bd, &g->bmc, has_grad, has_hess);
    transform_bvn_derivs(bd, &g->sig, J, has_hess);
    static const int sh_al[3] = {3, 4, 5}; /* gal_shape_alignment, 0-based into GalaxyPosParams */
    for (int u = 0; u < 2; ++u) fs1m->d[u] += f * bd->bvn_u_d[u];
    for (int g3 = 0; g3 < 3; ++g3) fs1m->d[sh_al[g3]] += f * bd->bvn_s_d[g3];
    fs1m->d[2] += g->dir * bd->f_pre;
    if (!has_hess) return;
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b)
        H_(fs1m, sh_al[a], sh_al[b]) += f * (bd->bvn_ss_h[a + 3 * b] + bd->bvn_s_d[a] * bd->bvn_s_d[b]);
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b)
        H_(fs1m, a, b) += f * (bd->bvn_uu_h[a + 2 * b] + bd->bvn_u_d[a] * bd->bvn_u_d[b]);
    for (int u = 0; u < 2; ++u) for (int s = 0; s < 3; ++s) {
        H_(fs1m, u, sh_al[s]) += f * (bd->bvn_us_h[u + 2 * s] + bd->bvn_u_d[u] * bd->bvn_s_d[s]);
        H_(fs1m, sh_al[s], u) = H_(fs1m, u, sh_al[s]);
    }
    for (int u = 0; u < 2; ++u) {
        H_(fs1m, u, 2) += bd->f_pre * g->dir * bd->bvn_u_d[u];
        H_(fs1m, 2, u) = H_(fs1m, u, 2);
    }
    for (int s = 0; s < 3; ++s) {
        H_(fs1m, sh_al[s], 2) += bd->f_pre * g->dir * bd->bvn_s_d[s];
        H_(fs1m, 2, sh_al[s]) = H_(fs1m, sh_al[s], 2);
    }
}

/* populate_gal_fsm! (fsm_util.jl:194-219) */
static void populate_gal_fsm(SF *fs1m, BvnDerivs *bd, const GalComp *mcs, int psf_K, int h, int w,
                             int active, const double J[4], int has_grad, int has_hess) {
    sf_zero(fs1m);
    double x[2] = {(double)h, (double)w};
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 8; ++j) {
        if (i == 0 || j < 6)
            for (int k = 0; k < psf_K; ++k)
                accum_galaxy_pos(fs1m, bd, &mcs[k + psf_K * (j + 8 * i)], x, J, active, has_grad, has_hess);
    }
}

/* ---- star density (fsm_util.jl:221-248; imaged_sources.jl:97-107) ---------- */
static double softpluslike(double x) { return 1000 * x > 1 ? 1000 * x - 1 : log(1000 * x); }

/* Dense Gaussian elimination with partial pivoting, n <= 53: solves M c = r in place */
static void solve_dense(int n, double *M, double *r) {
    for (int c = 0; c < n; ++c) {
        int piv = c; double best = fabs(M[c * n + c]);
        for (int q = c + 1; q < n; ++q) if (fabs(M[q * n + c]) > best) { best = fabs(M[q * n + c]); piv = q; }
        if (piv != c) {
            for (int k = 0; k < n; ++k) { double t = M[c * n + k]; M[c * n + k] = M[piv * n + k]; M[piv * n + k] = t; }
            double t = r[c]; r[c] = r[piv]; r[piv] = t;
        }
        for (int q = c + 1; q < n; ++q) {
            double f = M[q * n + c] / M[c * n + c];
            if (f == 0) continue;
            for (int k = c; k < n; ++k) M[q * n + k] -= f * M[c * n + k];
            r[q] -= f * r[c];
        }
    }
    for (int c = n - 1; c >= 0; --c) {
        double s = r[c];
        for (int k = c + 1; k < n; ++k) s -= M[c * n + k] * r[k];
        r[c] = s / M[c * n + c];
    }
}

/* Interpolations.jl `interpolate(A, BSpline(Cubic(Line())), OnGrid())` prefilter:
 * pad by one coefficient per side; interior rows [1/6 2/3 1/6] c = data;
 * boundary rows c[0] - 2 c[1] + c[2] = 0 (natural spline).  Separable. */
static void prefilter_1d(int n, const double *data, double *coef) { /* n data -> n+2 coefs */
    int m = n + 2;
    double *M = (double *)calloc((size_t)m * m, sizeof(double));
    double *r = (double *)calloc((size_t)m, sizeof(double));
    M[0] = 1; M[1] = -2; M[2] = 1; r[0] = 0;
    for (int q = 1; q <= n; ++q) { M[q * m + q - 1] = 1.0 / 6; M[q * m + q] = 2.0 / 3; M[q * m + q + 1] = 1.0 / 6; r[q] = data[q - 1]; }
    M[(m - 1) * m + m - 3] = 1; M[(m - 1) * m + m - 2] = -2; M[(m - 1) * m + m - 1] = 1; r[m - 1] = 0;
    solve_dense(m, M, r);
    memcpy(coef, r, sizeof(double) * m);
    free(M); free(r);
}

/* ImagePatch ctor lines 97-107: condition the raw stamp then prefilter. coef53 col-major 53x53 */
void celeste_oracle_spline_coefs(const double *stamp51, double *coef53) {
    enum { N = 51, C = 53 };
    double g[N * N]; double sum = 0;
    for (int k = 0; k < N * N; ++k) { g[k] = fmax(stamp51[k], 0.0); g[k] += 1e-6; }
    for (int k = 0; k < N * N; ++k) sum += g[k];
    for (int k = 0; k < N * N; ++k) g[k] = softpluslike(g[k] / sum);
    /* along dim 1 (rows of each column) */
    double tmp[C * N];
    for (int w = 0; w < N; ++w) prefilter_1d(N, g + N * w, tmp + C * w);
    /* along dim 2 */
    double line[N], cl[C];
    for (int h = 0; h < C; ++h) {
        for (int w = 0; w < N; ++w) line[w] = tmp[h + C * w];
        prefilter_1d(N, line, cl);
        for (int w = 0; w < C; ++w) coef53[h + C * w] = cl[w];
    }
}

/* cubic B-spline weights and their derivatives at fractional offset fx */
static void bs_weights(double fx, double w[4], double dw[4], double ddw[4]) {
    double o = 1 - fx;
    w[0] = o * o * o / 6; w[1] = 2.0 / 3 - fx * fx + fx * fx * fx / 2;
    w[2] = 2.0 / 3 - o * o + o * o * o / 2; w[3] = fx * fx * fx / 6;
    dw[0] = -o * o / 2; dw[1] = -2 * fx + 1.5 * fx * fx; dw[2] = 2 * o - 1.5 * o * o; dw[3] = fx * fx / 2;
    ddw[0] = o; ddw[1] = -2 + 3 * fx; ddw[2] = -2 + 3 * o; ddw[3] = fx;
}

/* itp[x, y] with value, gradient and Hessian wrt (x, y); 1-based coordinates on the 51-grid */
static void spline_eval(const double *coef53, double x, double y, double *S, double g[2], double hh[3]) {
    enum { C = 53 };
    int ix = (int)floor(x); if (ix < 1) ix = 1; if (ix > 50) ix = 50;
    int iy = (int)floor(y); if (iy < 1) iy = 1; if (iy > 50) iy = 50;
    double fx = x - ix, fy = y - iy;
    double wx[4], dwx[4], ddwx[4], wy[4], dwy[4], ddwy[4];
    bs_weights(fx, wx, dwx, ddwx); bs_weights(fy, wy, dwy, ddwy);
    /* padded 0-based index of coefficient "ix-1" (1-based unpadded) is ix-1 */
    double s = 0, sx = 0, sy = 0, sxx = 0, sxy = 0, syy = 0;
    for (int b = 0; b < 4; ++b) for (int a = 0; a < 4; ++a) {
        double c = coef53[(ix - 1 + a) + C * (iy - 1 + b)];
        s += c * wx[a] * wy[b]; sx += c * dwx[a] * wy[b]; sy += c * wx[a] * dwy[b];
        sxx += c * ddwx[a] * wy[b]; sxy += c * dwx[a] * dwy[b]; syy += c * wx[a] * ddwy[b];
    }
    *S = s; g[0] = sx; g[1] = sy; hh[0] = sxx; hh[1] = sxy; hh[2] = syy;
}
double celeste_oracle_spline_value(const double *coef53, double x, double y) {
    double s, g[2], hh[3]; spline_eval(coef53, x, y, &s, g, hh); return s;
}

/* star_light_density! (fsm_util.jl:225-248): fs0m has p = 2.  ForwardDiff
 * gradient/hessian of the closure == exact chain rule through m_pos = J (pos - wc) + pc */
static void star_light_density(SF *fs0m, const celeste_patch_t *p, const double *coef53, int h, int w,
                               const double pos[2], int active, int has_grad, int has_hess) {
    double m[2];
    linear_world_to_pix(p->wcs_jacobian, p->world_center, p->pixel_center, pos, m);
    double y, gy[2], hy[3];
    spline_eval(coef53, h - m[0] + 26, w - m[1] + 26, &y, gy, hy);
    double gv, g1, g2; /* softpluslikeinv and derivatives */
    if (y < 0) { gv = 1e-3 * exp(y); g1 = gv; g2 = gv; }
    else { gv = 1e-3 * (y + 1); g1 = 1e-3; g2 = 0; }
    fs0m->v = gv;
    if (!(active && has_grad)) return;
    const double *J = p->wcs_jacobian;
    /* d(idx_a)/d(pos_b) = -J[a,b] */
    double dy[2] = {-(gy[0] * J[0] + gy[1] * J[1]), -(gy[0] * J[2] + gy[1] * J[3])};
    fs0m->d[0] = g1 * dy[0]; fs0m->d[1] = g1 * dy[1];
    if (!has_hess) return;
    double Hs[4] = {hy[0], hy[1], hy[1], hy[2]};
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) {
        double q = 0;
        for (int c = 0; c < 2; ++c) for (int e = 0; e < 2; ++e) q += J[c + 2 * a] * Hs[c + 2 * e] * J[e + 2 * b];
        H_(fs0m, a, b) = g2 * dy[a] * dy[b] + g1 * q;
    }
}

/* ---- ElboIntermediateVariables (elbo_args.jl:29-113) ------------------------ */
typedef struct {
    SF fs0m, fs1m, E_G_s, E_G2_s, var_G_s, E_G, var_G, elbo_log_term, elbo;
    double ss_E[2][36], ss_E2[2][36], uu_E[2][4], uu_E2[2][4]; /* HessianSubmatrices */
    int64_t active_px, inactive_px;
    int has_grad, has_hess;
} ElboVars;

/* calculate_G_s! (elbo_objective.jl:17-233) */
static void calculate_G_s(const double *vs, ElboVars *ev, const SourceBrightness *sb, int b, int active) {
    SF *E_G_s = &ev->E_G_s, *E_G2_s = &ev->E_G2_s, *var_G_s = &ev->var_G_s;
    if (active) { sf_zero(E_G_s); sf_zero(E_G2_s); sf_zero(var_G_s); }
    else { E_G_s->v = 0; E_G2_s->v = 0; var_G_s->v = 0; }
    for (int i = 0; i < 2; ++i) {
        const SF *fsm = (i == 0) ? &ev->fs0m : &ev->fs1m;
        double a_i = vs[ID_IS_STAR + i];
        const SF *El = &sb->E_l_a[b][i], *Ell = &sb->E_ll_a[b][i];
        double fv = fsm->v, lv = El->v, llv = Ell->v;
        double lf = lv * fv, llff = llv * (fv * fv);
        E_G_s->v += a_i * lf; E_G2_s->v += a_i * llff;
        if (!(active && ev->has_grad)) continue;
        int ia = ID_IS_STAR + i;
        E_G_s->d[ia] += lf; E_G2_s->d[ia] += llff;
        int nshape = (i == 0) ? 2 : 6; /* shape_standard_alignment: canonical ids 0..nshape-1 */
        int pb[10]; bright_ids(i, pb);
        double tmp1 = lv * a_i, tmp2 = llv * 2 * fv * a_i;
        for (int s = 0; s < nshape; ++s) { E_G_s->d[s] += tmp1 * fsm->d[s]; E_G2_s->d[s] += tmp2 * fsm->d[s]; }
        for (int q = 0; q < 10; ++q) {
            E_G_s->d[pb[q]] = a_i * fv * El->d[q];
            E_G2_s->d[pb[q]] = a_i * (fv * fv) * Ell->d[q];
        }
        if (!ev->has_hess) continue;
        for (int q1 = 0; q1 < 10; ++q1) for (int q2 = 0; q2 < 10; ++q2) {
            H_(E_G_s, pb[q1], pb[q2]) = a_i * H_(El, q1, q2) * fv;
            H_(E_G2_s, pb[q1], pb[q2]) = (fv * fv) * a_i * H_(Ell, q1, q2);
        }
        double *ssE = ev->ss_E[i], *ssE2 = ev->ss_E2[i];
        for (int s1 = 0; s1 < nshape; ++s1) for (int s2 = 0; s2 < nshape; ++s2) {
            ssE[s1 + nshape * s2] = a_i * lv * H_(fsm, s1, s2);
            ssE2[s1 + nshape * s2] = 2 * a_i * llv * (fv * H_(fsm, s1, s2) + fsm->d[s1] * fsm->d[s2]);
        }
        for (int s1 = 0; s1 < nshape; ++s1) for (int s2 = 0; s2 < nshape; ++s2) {
            H_(E_G_s, s1, s2) = a_i * lv * H_(fsm, s1, s2);
            H_(E_G2_s, s1, s2) = ssE2[s1 + nshape * s2];
        }
        for (int u1 = 0; u1 < 2; ++u1) for (int u2 = 0; u2 < 2; ++u2) {
            ev->uu_E[i][u1 + 2 * u2] = ssE[u1 + nshape * u2];
            ev->uu_E2[i][u1 + 2 * u2] = ssE2[u1 + nshape * u2];
        }
        for (int q = 0; q < 10; ++q) {
            H_(E_G_s, pb[q], ia) = fv * El->d[q];
            H_(E_G2_s, pb[q], ia) = (fv * fv) * Ell->d[q];
            H_(E_G_s, ia, pb[q]) = H_(E_G_s, pb[q], ia);
            H_(E_G2_s, ia, pb[q]) = H_(E_G2_s, pb[q], ia);
        }
        for (int s = 0; s < nshape; ++s) {
            H_(E_G_s, s, ia) = lv * fsm->d[s];
            H_(E_G2_s, s, ia) = llv * 2 * fv * fsm->d[s];
            H_(E_G_s, ia, s) = H_(E_G_s, s, ia);
            H_(E_G2_s, ia, s) = H_(E_G2_s, s, ia);
        }
        for (int q = 0; q < 10; ++q) for (int s = 0; s < nshape; ++s) {
            H_(E_G_s, pb[q], s) = a_i * El->d[q] * fsm->d[s];
            H_(E_G2_s, pb[q], s) = 2 * a_i * Ell->d[q] * fv * fsm->d[s];
            H_(E_G_s, s, pb[q]) = H_(E_G_s, pb[q], s);
            H_(E_G2_s, s, pb[q]) = H_(E_G2_s, pb[q], s);
        }
    }
    if (active && ev->has_grad && ev->has_hess) {
        for (int u1 = 0; u1 < 2; ++u1) for (int u2 = 0; u2 < 2; ++u2) {
            H_(E_G_s, u1, u2) = ev->uu_E[0][u1 + 2 * u2] + ev->uu_E[1][u1 + 2 * u2];
            H_(E_G2_s, u1, u2) = ev->uu_E2[0][u1 + 2 * u2] + ev->uu_E2[1][u1 + 2 * u2];
        }
    }
    var_G_s->v = E_G2_s->v - (E_G_s->v * E_G_s->v);
    if (!(active && ev->has_grad)) return;
    for (int k = 0; k < P; ++k) var_G_s->d[k] = E_G2_s->d[k] - 2 * E_G_s->v * E_G_s->d[k];
    if (!ev->has_hess) return;
    for (int i2 = 0; i2 < P; ++i2) for (int i1 = 0; i1 <= i2; ++i1) {
        H_(var_G_s, i1, i2) = H_(E_G2_s, i1, i2) - 2 * (E_G_s->v * H_(E_G_s, i1, i2) + E_G_s->d[i1] * E_G_s->d[i2]);
        H_(var_G_s, i2, i1) = H_(var_G_s, i1, i2);
    }
}

/* add_sources_sf! (SensitiveFloats.jl:215-250): the single-source SF goes into block `sa` of the all-sources SF */
static void add_sources_sf(SF *all, const SF *s, int sa, int has_grad, int has_hess) {
    all->v += s->v;
    const int o = P * sa;
    if (has_grad) for (int k = 0; k < P; ++k) all->d[o + k] = all->d[o + k] + s->d[k];
    if (has_hess) for (int j = 0; j < P; ++j) for (int i = 0; i < P; ++i) H_(all, o + i, o + j) += s->h[i + P * j];
}

/* add_elbo_log_term! (elbo_objective.jl:274-327) */
static void add_elbo_log_term(ElboVars *ev, float x_nbm, float iota) {
    double E = ev->E_G.v, V = ev->var_G.v;
    double log_term = log(E) - V / (2.0 * (E * E));
    /* log(iota) is evaluated on a Float32 argument in the reference */
    double log_iota = (double)(float)log((double)iota);
    ev->elbo.v += (double)x_nbm * (log_iota + log_term);
    if (!ev->has_grad) return;
    double g_d[2] = {-0.5 / (E * E), 1 / E + V / (E * E * E)};
    double g_h[4] = {0, 1 / (E * E * E), 1 / (E * E * E), -(1 / (E * E) + 3 * V / (E * E * E * E))};
    combine_sfs(&ev->var_G, &ev->E_G, &ev->elbo_log_term, log_term, g_d, g_h, 1, ev->has_hess);
    const int pt = ev->elbo.p;
    for (int k = 0; k < pt; ++k) ev->elbo.d[k] += (double)x_nbm * ev->elbo_log_term.d[k];
    if (ev->has_hess) for (int k = 0; k < pt * pt; ++k) ev->elbo.h[k] += (double)x_nbm * ev->elbo_log_term.h[k];
}

/* add_scaled_sfs! (SensitiveFloats.jl:185-208) */
static void add_scaled_sfs(SF *a, const SF *b, double scale, int has_grad, int has_hess) {
    a->v += scale * b->v;
    const int pt = a->p;
    if (has_grad) for (int k = 0; k < pt; ++k) a->d[k] += scale * b->d[k];
    if (has_hess) for (int i2 = 0; i2 < pt; ++i2) for (int i1 = 0; i1 <= i2; ++i1) {
        H_(a, i1, i2) += scale * H_(b, i1, i2);
        H_(a, i2, i1) = H_(a, i1, i2);
    }
}

static int patch_bitmap(const celeste_problem_t *pr, int n, const celeste_patch_t *p, int h2, int w2) {
    /* 1-based h2, w2 */
    if (p->bitmap) return p->bitmap[(h2 - 1) + (size_t)p->H2 * (w2 - 1)] != 0;
    const celeste_image_t *im = &pr->images[n];
    float x = im->pixels[(p->off_h + h2 - 1) + (size_t)im->H * (p->off_w + w2 - 1)];
    return !isnan(x); /* imaged_sources.jl:94-95 */
}

/* ---- KL (src/deterministic_vi/elbo_kl.jl:94-154); analytic derivatives ---- */
static void inv4_logdet(const double *S /*col-major 4x4*/, double *Inv, double *logdet) {
    double a[4][8];
    for (int r = 0; r < 4; ++r) { for (int c = 0; c < 4; ++c) { a[r][c] = S[r + 4 * c]; a[r][4 + c] = (r == c); } }
    double det = 1;
    for (int c = 0; c < 4; ++c) {
        int piv = c; for (int r = c + 1; r < 4; ++r) if (fabs(a[r][c]) > fabs(a[piv][c])) piv = r;
        if (piv != c) { for (int k = 0; k < 8; ++k) { double t = a[c][k]; a[c][k] = a[piv][k]; a[piv][k] = t; } det = -det; }
        det *= a[c][c];
        double inv = 1 / a[c][c];
        for (int k = 0; k < 8; ++k) a[c][k] *= inv;
        for (int r = 0; r < 4; ++r) if (r != c) { double f = a[r][c]; for (int k = 0; k < 8; ++k) a[r][k] -= f * a[c][k]; }
    }
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) Inv[r + 4 * c] = a[r][4 + c];
    *logdet = log(det);
}

/* value of subtract_kl(vs) and, if d/h non-NULL, its gradient / Hessian (44, 44x44 col-major) */
void celeste_oracle_subtract_kl(const celeste_prior_t *prior, const double *vs, double *v, double *d, double *h) {
    if (!prior) prior = &ORACLE_DEFAULT_PRIOR;
    double kl = 0;
    double D[P]; double *Hm = (double *)calloc(P * P, sizeof(double));
    memset(D, 0, sizeof D);
#define HH(i, j) Hm[(i) + P * (j)]
#define HS(i, j, val) do { double v__ = (val); HH(i, j) += v__; if ((i) != (j)) HH(j, i) += v__; } while (0)
    /* kl_source_a: categorical_kl(vs[is_star], prior.is_star) */
    for (int i = 0; i < 2; ++i) {
        double a = vs[ID_IS_STAR + i];
        double t = log(a) - log(prior->is_star[i]);
        kl -= a * t;
        D[ID_IS_STAR + i] -= t + 1;
        HS(ID_IS_STAR + i, ID_IS_STAR + i, -1 / a);
    }
    for (int i = 0; i < 2; ++i) {
        int ia = ID_IS_STAR + i;
        double a = vs[ia];
        /* kl_source_k */
        double ck = 0;
        for (int dd = 0; dd < 8; ++dd) {
            int ik = ID_K + 8 * i + dd;
            double k = vs[ik];
            double t = log(k) - log(prior->k[i][dd]);
            ck += k * t;
            D[ik] -= a * (t + 1);
            HS(ik, ik, -a / k);
            HS(ia, ik, -(t + 1));
        }
        kl -= a * ck; D[ia] -= ck;
        /* kl_source_r: gaussian_kl(flux_loc, flux_scale, mean, var) */
        {
            int ir = ID_FLUX_LOC + i, iv = ID_FLUX_SCALE + i;
            double mu1 = vs[ir], var1 = vs[iv], mu2 = prior->flux_mean[i], var2 = prior->flux_var[i];
            double g = .5 * (log(var2) - log(var1) + (var1 + (mu1 - mu2) * (mu1 - mu2)) / var2 - 1);
            double g_r = (mu1 - mu2) / var2, g_v = .5 * (-1 / var1 + 1 / var2);
            kl -= a * g; D[ia] -= g; D[ir] -= a * g_r; D[iv] -= a * g_v;
            HS(ia, ir, -g_r); HS(ia, iv, -g_v);
            HS(ir, ir, -a / var2); HS(iv, iv, -a * .5 / (var1 * var1));
        }
        /* kl_source_c: sum_d a k_d diagmvn_mvn_kl(color_mean, color_var, mu_d, Sigma_d) */
        for (int dd = 0; dd < 8; ++dd) {
            int ik = ID_K + 8 * i + dd;
            double k = vs[ik];
            double Inv[16], logdet;
            inv4_logdet(prior->color_cov[i][dd], Inv, &logdet);
            double diff[4], m = 0, sl = 0, tr = 0;
            for (int c = 0; c < 4; ++c) diff[c] = prior->color_mean[i][dd][c] - vs[ID_COLOR_MEAN + 4 * i + c];
            for (int c = 0; c < 4; ++c) { tr += Inv[c + 4 * c] * vs[ID_COLOR_VAR + 4 * i + c]; sl += log(vs[ID_COLOR_VAR + 4 * i + c]); }
            double quad = 0, Ld[4];
            for (int r = 0; r < 4; ++r) { Ld[r] = 0; for (int c = 0; c < 4; ++c) Ld[r] += Inv[r + 4 * c] * diff[c]; quad += diff[r] * Ld[r]; }
            m = 0.5 * ((tr - 4) + quad + (logdet - sl));
            kl -= a * k * m; D[ia] -= k * m; D[ik] -= a * m;
            HS(ia, ik, -m);
            for (int c = 0; c < 4; ++c) {
                int im = ID_COLOR_MEAN + 4 * i + c, il = ID_COLOR_VAR + 4 * i + c;
                double lam = vs[il];
                double m_mu = -Ld[c];                  /* dm/dmu1_c */
                double m_l = 0.5 * (Inv[c + 4 * c] - 1 / lam);
                D[im] -= a * k * m_mu; D[il] -= a * k * m_l;
                HS(ia, im, -k * m_mu); HS(ia, il, -k * m_l);
                HS(ik, im, -a * m_mu); HS(ik, il, -a * m_l);
                HS(il, il, -a * k * 0.5 / (lam * lam));
                for (int c2 = 0; c2 <= c; ++c2) {
                    int im2 = ID_COLOR_MEAN + 4 * i + c2;
                    HS(im2, im, -a * k * Inv[c2 + 4 * c]);
                }
            }
        }
    }
    /* source_e_log_prob */
    {
        double x = vs[ID_RADIUS], mu = prior->gal_radius_px_mean, s2 = prior->gal_radius_px_var;
        kl += -0.5 * (log(2 * M_PI) + log(s2) + (x - mu) * (x - mu) / s2);
        D[ID_RADIUS] += -(x - mu) / s2;
        HS(ID_RADIUS, ID_RADIUS, -1 / s2);
    }
    *v = kl;
    if (d) memcpy(d, D, sizeof D);
    if (h) memcpy(h, Hm, sizeof(double) * P * P);
    free(Hm);
#undef HH
#undef HS
}

/* standalone KL closed forms for the known-answer tests (elbo_kl.jl:24-84) */
double celeste_oracle_categorical_kl(const double *p1, const double *p2, int n) {
    double kl = 0; for (int i = 0; i < n; ++i) kl += p1[i] * (log(p1[i]) - log(p2[i])); return kl;
}
double celeste_oracle_gaussian_kl(double mu1, double var1, double mu2, double var2) {
    return .5 * (log(var2) - log(var1) + (var1 + (mu1 - mu2) * (mu1 - mu2)) / var2 - 1);
}

/* PSF.get_psf_at_point (src/PSF.jl:150-161) for one point */
double celeste_oracle_psf_at_point(const double *psf, int K, double row, double col) {
    double s = 0;
    for (int k = 0; k < K; ++k) {
        const double *pc = psf + 6 * k;
        double x0 = row - pc[1], x1 = col - pc[2];
        double det = pc[3] * pc[5] - pc[4] * pc[4];
        double i11 = pc[5] / det, i12 = -pc[4] / det, i22 = pc[3] / det;
        double q = x0 * (i11 * x0 + i12 * x1) + x1 * (i12 * x0 + i22 * x1);
        s += pc[0] * exp(-0.5 * q - 0.5 * log(det)) / (2 * M_PI);
    }
    return s;
}

/* ---- elbo_likelihood + elbo (elbo_objective.jl:400-492) ---------------------- */
/* coefs_all: n_stamps x 53 x 53 spline coefficients (ImagePatch.itp_psf is built
 * once per patch at construction time, not per elbo() call).
 * active[0..Sa-1]: ElboArgs.active_sources.  The local source list (ElboArgs.S) is the active sources followed by
 * every neighbour of an active source that is not itself active.  d: P x Sa, h: (P Sa) x (P Sa), col-major. */
static int oracle_elbo_multi(const celeste_problem_t *pr, const double *coefs_all, const double *vp, int Sa,
                             const int32_t *active_src, uint32_t flags, double *v, double *d, double *h,
                             int64_t *n_active_px, int64_t *n_inactive_px) {
    if (!pr || !vp || Sa < 1 || !active_src) return CELESTE_ERR_INVALID_ARG;
    for (int a = 0; a < Sa; ++a) {
        if (active_src[a] < 0 || active_src[a] >= pr->n_sources) return CELESTE_ERR_INVALID_ARG;
        for (int a2 = 0; a2 < a; ++a2) if (active_src[a2] == active_src[a]) return CELESTE_ERR_INVALID_ARG;
    }
    const int N = pr->n_images, K = pr->psf_K;
    const int has_hess = (flags & CELESTE_FLAG_HESS) != 0;
    const int has_grad = has_hess || (flags & CELESTE_FLAG_GRAD) != 0;
    const int PT = P * Sa;
    /* local source list: [active...; neighbours of the active sources] */
    int cap = Sa;
    for (int a = 0; a < Sa; ++a)
        if (pr->nbr_offsets) cap += (int)(pr->nbr_offsets[active_src[a] + 1] - pr->nbr_offsets[active_src[a]]);
    int *src = (int *)malloc(sizeof(int) * cap);
    int S = 0;
    for (int a = 0; a < Sa; ++a) src[S++] = active_src[a];
    for (int a = 0; a < Sa && pr->nbr_offsets; ++a)
        for (int64_t q = pr->nbr_offsets[active_src[a]]; q < pr->nbr_offsets[active_src[a] + 1]; ++q) {
            const int c = pr->nbr_index[q];
            int seen = 0;
            for (int k = 0; k < S; ++k) if (src[k] == c) { seen = 1; break; }
            if (!seen) src[S++] = c;
        }
    for (int q = 0; q < S; ++q) for (int k = 0; k < P; ++k)
        if (!isfinite(vp[(size_t)src[q] * P + k])) { free(src); return CELESTE_ERR_NONFINITE_INPUT; }

    ElboVars ev; memset(&ev, 0, sizeof ev);
    ev.has_grad = has_grad; ev.has_hess = has_hess;
    ev.fs0m = sf_new(2); ev.fs1m = sf_new(6);
    ev.E_G_s = sf_new(P); ev.E_G2_s = sf_new(P); ev.var_G_s = sf_new(P);
    ev.E_G = sf_new(PT); ev.var_G = sf_new(PT); ev.elbo_log_term = sf_new(PT); ev.elbo = sf_new(PT);

    SourceBrightness *sbs = (SourceBrightness *)malloc(sizeof(SourceBrightness) * S);
    for (int q = 0; q < S; ++q) sb_load(&sbs[q], vp + (size_t)src[q] * P, q < Sa && has_grad);
    GalComp *mcs = (GalComp *)malloc(sizeof(GalComp) * S * K * 16);
    const double **coefs = (const double **)malloc(sizeof(double *) * S);
    BvnDerivs bd; memset(&bd, 0, sizeof bd);

    for (int n = 0; n < N; ++n) {
        const celeste_image_t *img = &pr->images[n];
        const int b = img->band - 1;
        for (int q = 0; q < S; ++q) {
            const celeste_patch_t *p = oracle_patch_at(pr, src[q], n);
            load_bvn_mixtures_source(mcs + (size_t)q * K * 16, p, K, vp + (size_t)src[q] * P,
                                     has_grad && q < Sa, has_hess);
            coefs[q] = coefs_all + (size_t)p->stamp * 53 * 53;
        }
        /* pixels of several active patches are visited once (elbo_objective.jl:430-470) */
        unsigned char *visited = Sa > 1 ? (unsigned char *)calloc((size_t)img->H * img->W, 1) : NULL;
        for (int sa = 0; sa < Sa; ++sa) {
        const celeste_patch_t *pa = oracle_patch_at(pr, src[sa], n);
        for (int w2 = 1; w2 <= pa->W2; ++w2) for (int h2 = 1; h2 <= pa->H2; ++h2) {
            int hh = pa->off_h + h2, ww = pa->off_w + w2; /* 1-based image coords */
            if (!patch_bitmap(pr, n, pa, h2, w2)) continue;
            if (visited) {
                unsigned char *vis = &visited[(hh - 1) + (size_t)img->H * (ww - 1)];
                if (*vis) continue;
                *vis = 1;
            }
            float x_nbm = img->pixels[(hh - 1) + (size_t)img->H * (ww - 1)];
            if (isnan(x_nbm)) continue;
            /* add_pixel_term! (elbo_objective.jl:330-392) */
            sf_zero(&ev.E_G); sf_zero(&ev.var_G);
            for (int q = 0; q < S; ++q) {
                const celeste_patch_t *p = oracle_patch_at(pr, src[q], n);
                int ph2 = hh - p->off_h, pw2 = ww - p->off_w;
                if (!(1 <= ph2 && ph2 <= p->H2 && 1 <= pw2 && pw2 < p->W2)) continue;
                if (!patch_bitmap(pr, n, p, ph2, pw2)) continue;
                int active = (q < Sa);
                if (active) ev.active_px++; else ev.inactive_px++;
                const double *vs = vp + (size_t)src[q] * P;
                star_light_density(&ev.fs0m, p, coefs[q], hh, ww, vs + ID_POS, active, has_grad, has_hess);
                populate_gal_fsm(&ev.fs1m, &bd, mcs + (size_t)q * K * 16, K, hh, ww, active, p->wcs_jacobian, has_grad, has_hess);
                /* accumulate_source_pixel_brightness! (elbo_objective.jl:240-259) */
                calculate_G_s(vs, &ev, &sbs[q], b, active);
                if (active) {
                    add_sources_sf(&ev.E_G, &ev.E_G_s, q, has_grad, has_hess);
                    add_sources_sf(&ev.var_G, &ev.var_G_s, q, has_grad, has_hess);
                } else { ev.E_G.v += ev.E_G_s.v; ev.var_G.v += ev.var_G_s.v; }
            }
            ev.E_G.v += (double)img->sky[(hh - 1) + (size_t)img->H * (ww - 1)];
            float iota = img->nelec_per_nmgy[hh - 1];
            add_elbo_log_term(&ev, x_nbm, iota);
            add_scaled_sfs(&ev.elbo, &ev.E_G, -(double)iota, has_grad, has_hess);
            ev.elbo.v -= lgamma((double)x_nbm + 1.0);
        }
        }
        free(visited);
    }
    if (flags & CELESTE_FLAG_KL) { /* subtract_kl_all_sources! (elbo_kl.jl:214-225) */
        double kv; double *kd = (double *)malloc(sizeof(double) * P), *kh = (double *)malloc(sizeof(double) * P * P);
        for (int sa = 0; sa < Sa; ++sa) {
            celeste_oracle_subtract_kl(pr->prior, vp + (size_t)src[sa] * P, &kv, kd, kh);
            ev.elbo.v += kv;
            if (has_grad) for (int k = 0; k < P; ++k) ev.elbo.d[P * sa + k] += kd[k];
            if (has_hess) for (int j = 0; j < P; ++j) for (int i = 0; i < P; ++i) H_(&ev.elbo, P * sa + i, P * sa + j) += kh[i + P * j];
        }
        free(kd); free(kh);
    }
    int status = CELESTE_OK;
    if (!isfinite(ev.elbo.v)) status = CELESTE_ERR_NONFINITE_RESULT;
    if (has_grad) for (int k = 0; k < PT; ++k) if (!isfinite(ev.elbo.d[k])) status = CELESTE_ERR_NONFINITE_RESULT;
    if (has_hess) for (int k = 0; k < PT * PT; ++k) if (!isfinite(ev.elbo.h[k])) status = CELESTE_ERR_NONFINITE_RESULT;
    if (v) *v = ev.elbo.v;
    if (d && has_grad) memcpy(d, ev.elbo.d, sizeof(double) * PT);
    if (h && has_hess) memcpy(h, ev.elbo.h, sizeof(double) * PT * PT);
    if (n_active_px) *n_active_px = ev.active_px;
    if (n_inactive_px) *n_inactive_px = ev.inactive_px;

    for (int q = 0; q < S; ++q) sb_free(&sbs[q]);
    free(sbs); free(mcs); free((void *)coefs); free(src);
    sf_free(&ev.fs0m); sf_free(&ev.fs1m); sf_free(&ev.E_G_s); sf_free(&ev.E_G2_s); sf_free(&ev.var_G_s);
    sf_free(&ev.E_G); sf_free(&ev.var_G); sf_free(&ev.elbo_log_term); sf_free(&ev.elbo);
    return status;
}

static int oracle_elbo_impl(const celeste_problem_t *pr, const double *coefs_all, const double *vp, int32_t target,
                            uint32_t flags, double *v, double *d, double *h, int64_t *n_active_px,
                            int64_t *n_inactive_px) {
    if (!pr || !vp || target < 0 || target >= pr->n_sources) return CELESTE_ERR_INVALID_ARG;
    return oracle_elbo_multi(pr, coefs_all, vp, 1, &target, flags, v, d, h, n_active_px, n_inactive_px);
}

static double *all_coefs(const celeste_problem_t *pr) {
    double *c = (double *)malloc(sizeof(double) * (size_t)(pr->n_stamps > 0 ? pr->n_stamps : 1) * 53 * 53);
    for (int k = 0; k < pr->n_stamps; ++k) celeste_oracle_spline_coefs(pr->stamps + (size_t)k * 51 * 51, c + (size_t)k * 53 * 53);
    return c;
}

int celeste_oracle_elbo(const celeste_problem_t *pr, const double *vp, int32_t target, uint32_t flags,
                        double *v, double *d, double *h, int64_t *n_active_px, int64_t *n_inactive_px) {
    if (!pr) return CELESTE_ERR_INVALID_ARG;
    double *c = all_coefs(pr);
    int st = oracle_elbo_impl(pr, c, vp, target, flags, v, d, h, n_active_px, n_inactive_px);
    free(c);
    return st;
}

/* elbo() with several active sources (ElboArgs.active_sources, Sa >= 1): d is P x Sa, h (P Sa) x (P Sa) */
int celeste_oracle_elbo_multi(const celeste_problem_t *pr, const double *vp, int32_t n_active, const int32_t *active,
                              uint32_t flags, double *v, double *d, double *h, int64_t *n_active_px,
                              int64_t *n_inactive_px) {
    if (!pr) return CELESTE_ERR_INVALID_ARG;
    double *c = all_coefs(pr);
    int st = oracle_elbo_multi(pr, c, vp, n_active, active, flags, v, d, h, n_active_px, n_inactive_px);
    free(c);
    return st;
}

/* Sweep over a list of targets, OpenMP dynamic schedule over sources: the
 * reference's spin-locked per-source work queue (ParallelRun.jl:546-607). */
int celeste_oracle_elbo_batch(const celeste_problem_t *pr, const double *vp, int32_t n_targets,
                              const int32_t *targets, uint32_t flags, double *v, double *d, double *h,
                              int64_t *counters, int32_t *status, int32_t n_threads) {
    int worst = CELESTE_OK;
    (void)n_threads;
    if (!pr) return CELESTE_ERR_INVALID_ARG;
    double *coefs_all = all_coefs(pr);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads > 0 ? n_threads : 1)
#endif
    for (int t = 0; t < n_targets; ++t) {
        int64_t na = 0, ni = 0;
        int st = oracle_elbo_impl(pr, coefs_all, vp, targets[t], flags, v ? v + t : NULL, d ? d + (size_t)t * P : NULL,
                                     h ? h + (size_t)t * P * P : NULL, &na, &ni);
        if (counters) { counters[2 * t] = na; counters[2 * t + 1] = ni; }
        if (status) status[t] = st;
        if (st != CELESTE_OK) {
#ifdef _OPENMP
#pragma omp critical
#endif
            worst = st;
        }
    }
    free(coefs_all);
    return worst;
}

/* ---- per-function hooks for the micro-goldens (tests/golden/micro/, tests/test_oracle_micro.py) ----------------
 * They expose the intermediate results of the restated functions so that each can be compared with an independent
 * 50-digit evaluation (mpmath; values and numerically differentiated derivatives).  Test infrastructure only. */

/* eval_bvn_pdf! + get_bvn_derivs! + GalaxySigmaDerivs + transform_bvn_derivs! for ONE component
 * (BivariateNormals.jl:143-572) with mean = `mean`, covariance = tau + nuBar XiXi(ratio, angle, radius), evaluated at x.
 * out[87]: f_pre, py1, py2 | bvn_x_d[2] bvn_sig_d[3] bvn_xx_h[4] bvn_xsig_h[6] bvn_sigsig_h[9] | j[9] t[27] |
 *          bvn_u_d[2] bvn_uu_h[4] bvn_s_d[3] bvn_ss_h[9] bvn_us_h[6] */
void celeste_oracle_micro_bvn(const double mean[2], const double tau[3], double weight, const double x[2],
                              const double J[4], double ratio, double angle, double radius, double nuBar, double *out) {
    double XiXi[4];
    celeste_oracle_get_bvn_cov(ratio, angle, radius, XiXi);
    double cov[4] = {tau[0] + nuBar * XiXi[0], tau[1] + nuBar * XiXi[1], tau[1] + nuBar * XiXi[2], tau[2] + nuBar * XiXi[3]};
    Bvn b; bvn_make(&b, mean, cov, weight, 1);
    SigSF s; sigsf_make(&s, angle, ratio, radius, XiXi, nuBar, 1);
    BvnDerivs bd; memset(&bd, 0, sizeof bd);
    eval_bvn_pdf(&bd, &b, x);
    get_bvn_derivs(&bd, &b, 1, 1);
    transform_bvn_derivs(&bd, &s, J, 1);
    int k = 0;
    out[k++] = bd.f_pre; out[k++] = bd.py1; out[k++] = bd.py2;
    for (int q = 0; q < 2; ++q) out[k++] = bd.bvn_x_d[q];
    for (int q = 0; q < 3; ++q) out[k++] = bd.bvn_sig_d[q];
    for (int q = 0; q < 4; ++q) out[k++] = bd.bvn_xx_h[q];
    for (int q = 0; q < 6; ++q) out[k++] = bd.bvn_xsig_h[q];
    for (int q = 0; q < 9; ++q) out[k++] = bd.bvn_sigsig_h[q];
    for (int q = 0; q < 9; ++q) out[k++] = s.j[q];
    for (int q = 0; q < 27; ++q) out[k++] = s.t[q];
    for (int q = 0; q < 2; ++q) out[k++] = bd.bvn_u_d[q];
    for (int q = 0; q < 4; ++q) out[k++] = bd.bvn_uu_h[q];
    for (int q = 0; q < 3; ++q) out[k++] = bd.bvn_s_d[q];
    for (int q = 0; q < 9; ++q) out[k++] = bd.bvn_ss_h[q];
    for (int q = 0; q < 6; ++q) out[k++] = bd.bvn_us_h[q];
}

/* SourceBrightness (source_brightness.jl:27-202): out[(b + 5 i) * 111 + ...] = v, d[10], h[100] of E_l_a[b, i], then the
 * same block again for E_ll_a */
void celeste_oracle_micro_brightness(const double *vs, double *out) {
    SourceBrightness sb; sb_load(&sb, vs, 1);
    for (int w = 0; w < 2; ++w) for (int i = 0; i < 2; ++i) for (int b = 0; b < NB; ++b) {
        const SF *E = w == 0 ? &sb.E_l_a[b][i] : &sb.E_ll_a[b][i];
        double *o = out + (size_t)((w * 2 + i) * NB + b) * 111;
        o[0] = E->v; memcpy(o + 1, E->d, sizeof(double) * 10); memcpy(o + 11, E->h, sizeof(double) * 100);
    }
    sb_free(&sb);
}

/* One pixel (h, w; 1-based) of image n for source s as the ACTIVE source, nothing else on the pixel:
 * star_light_density!, populate_gal_fsm!, calculate_G_s!, then E_G = sky + E_G_s, var_G = var_G_s and
 * add_elbo_log_term! (elbo_objective.jl:17-327; fsm_util.jl:194-248).
 * out: fs0m (1 + 2 + 4) | fs1m (1 + 6 + 36) | E_G_s (1 + 44 + 1936) | var_G_s (same) | elbo_log_term (same) */
int celeste_oracle_micro_pixel(const celeste_problem_t *pr, const double *vp, int32_t s, int32_t n, int32_t h, int32_t w,
                               double *out) {
    if (!pr || !vp || s < 0 || s >= pr->n_sources || n < 0 || n >= pr->n_images) return CELESTE_ERR_INVALID_ARG;
    const celeste_image_t *img = &pr->images[n];
    const celeste_patch_t *p = oracle_patch_at(pr, s, n);
    const int K = pr->psf_K;
    const double *vs = vp + (size_t)s * P;
    double *coef = (double *)malloc(sizeof(double) * 53 * 53);
    celeste_oracle_spline_coefs(pr->stamps + (size_t)p->stamp * 51 * 51, coef);
    ElboVars ev; memset(&ev, 0, sizeof ev);
    ev.has_grad = 1; ev.has_hess = 1;
    ev.fs0m = sf_new(2); ev.fs1m = sf_new(6);
    ev.E_G_s = sf_new(P); ev.E_G2_s = sf_new(P); ev.var_G_s = sf_new(P);
    ev.E_G = sf_new(P); ev.var_G = sf_new(P); ev.elbo_log_term = sf_new(P); ev.elbo = sf_new(P);
    SourceBrightness sb; sb_load(&sb, vs, 1);
    GalComp *mcs = (GalComp *)malloc(sizeof(GalComp) * K * 16);
    BvnDerivs bd; memset(&bd, 0, sizeof bd);
    load_bvn_mixtures_source(mcs, p, K, vs, 1, 1);
    star_light_density(&ev.fs0m, p, coef, h, w, vs + ID_POS, 1, 1, 1);
    populate_gal_fsm(&ev.fs1m, &bd, mcs, K, h, w, 1, p->wcs_jacobian, 1, 1);
    calculate_G_s(vs, &ev, &sb, img->band - 1, 1);
    add_sources_sf(&ev.E_G, &ev.E_G_s, 0, 1, 1);
    add_sources_sf(&ev.var_G, &ev.var_G_s, 0, 1, 1);
    ev.E_G.v += (double)img->sky[(h - 1) + (size_t)img->H * (w - 1)];
    add_elbo_log_term(&ev, 1.0f, img->nelec_per_nmgy[h - 1]);
    double *o = out;
    *o++ = ev.fs0m.v; memcpy(o, ev.fs0m.d, sizeof(double) * 2); o += 2; memcpy(o, ev.fs0m.h, sizeof(double) * 4); o += 4;
    *o++ = ev.fs1m.v; memcpy(o, ev.fs1m.d, sizeof(double) * 6); o += 6; memcpy(o, ev.fs1m.h, sizeof(double) * 36); o += 36;
    const SF *big[3] = {&ev.E_G_s, &ev.var_G_s, &ev.elbo_log_term};
    for (int q = 0; q < 3; ++q) {
        *o++ = big[q]->v; memcpy(o, big[q]->d, sizeof(double) * P); o += P; memcpy(o, big[q]->h, sizeof(double) * P * P); o += P * P;
    }
    sb_free(&sb); free(mcs); free(coef);
    sf_free(&ev.fs0m); sf_free(&ev.fs1m); sf_free(&ev.E_G_s); sf_free(&ev.E_G2_s); sf_free(&ev.var_G_s);
    sf_free(&ev.E_G); sf_free(&ev.var_G); sf_free(&ev.elbo_log_term); sf_free(&ev.elbo);
    return CELESTE_OK;
}
