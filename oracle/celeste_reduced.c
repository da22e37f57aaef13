/*
 * celeste_reduced.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (bench.py cpu_baseline leg and tests/ only).
 *
 * A second CPU evaluation of elbo() (elbo_objective.jl:400-492) that uses the *reduced-variable* algorithm of the
 * HIP engine instead of the reference's dense per-pixel SensitiveFloats (celeste_oracle.c): per band the pixel term
 * depends on 10 reduced variables z = (c0, c1, q0, q1 | m1, m2 | dev | Xi11, Xi12, Xi22); every pixel contributes a
 * 68-double record (value, 10 gradient, 55 packed Hessian entries, 2 counters) built from closed-form Gaussian
 * derivatives (Hermite polynomials), and one chain rule per (source, image) lifts the sums to the 44 parameters.
 * Purpose ("CPU baseline"): timed next to the dense restatement so that the algorithmic part of the
 * GPU/CPU ratio (dense -> reduced) and the hardware part (reduced CPU -> reduced GPU) can be read off separately.
 * It is checked against the dense oracle (tests/test_oracle_reduced.py), not against the device.
 * Plain C, fp64, libm exp; OpenMP over target sources.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/celeste_mi355x.h"
#include "patch_lookup.h"

#define P 44
#define ZV 10
#define NREC 68
#define REC_H0 11
#define REC_CNT 66
#define NP 28 /* parameters with likelihood derivatives (the k block only enters the KL) */
#define COEF 53

void celeste_oracle_galaxy_prototypes(double eta[16], double nu[16]);
void celeste_oracle_spline_coefs(const double *stamp51, double *coef53);
void celeste_oracle_subtract_kl(const celeste_prior_t *prior, const double *vs, double *v, double *d, double *h);

static inline int hidx(int i, int j) { return REC_H0 + ZV * i - i * (i - 1) / 2 + (j - i); } /* i <= j */

typedef struct { double p11, p12, p22, xi1, xi2, w0, wd, nu; } RComp;
typedef struct { double m1, m2, c0, c1, q0, q1; } RSrc;

static void r_bvn_cov(double ab, double angle, double scale, double *x11, double *x12, double *x22) {
    const double sp = sin(angle), cp = cos(angle), abt = ab * ab - 1.0, s2 = scale * scale;
    *x12 = -s2 * cp * sp * abt; *x11 = s2 * (1.0 + abt * sp * sp); *x22 = s2 * (1.0 + abt * cp * cp);
}

/* E[l_b | a = i], E[l_b^2 | a = i] (source_brightness.jl:46-50, 123-127); b 0-based */
static void r_brightness(const double *vs, int i, int b, double *El, double *Ell) {
    const double r = vs[6 + i], v = vs[8 + i];
    const double *cm = vs + 10 + 4 * i, *cv = vs + 18 + 4 * i;
    double e = exp(r + 0.5 * v), ee = exp(2 * r + 2 * v);
    if (b >= 3) { e *= exp(cm[2] + .5 * cv[2]); ee *= exp(2 * cm[2] + 2 * cv[2]); }
    if (b >= 4) { e *= exp(cm[3] + .5 * cv[3]); ee *= exp(2 * cm[3] + 2 * cv[3]); }
    if (b <= 1) { e *= exp(-cm[1] + .5 * cv[1]); ee *= exp(-2 * cm[1] + 2 * cv[1]); }
    if (b <= 0) { e *= exp(-cm[0] + .5 * cv[0]); ee *= exp(-2 * cm[0] + 2 * cv[0]); }
    *El = e; *Ell = ee;
}

/* load_bvn_mixtures! for one (source, image) in component-record form */
static void r_prep(const celeste_patch_t *p, int K, const double *vs, int band0, const double *eta, const double *nu,
                   RComp *comps, RSrc *si) {
    const double d0 = vs[0] - p->world_center[0], d1 = vs[1] - p->world_center[1];
    const double *J = p->wcs_jacobian;
    si->m1 = J[0] * d0 + J[2] * d1 + p->pixel_center[0];
    si->m2 = J[1] * d0 + J[3] * d1 + p->pixel_center[1];
    double x11, x12, x22;
    r_bvn_cov(vs[3], vs[4], vs[5], &x11, &x12, &x22);
    int c = 0;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < (i == 0 ? 8 : 6); ++j)
            for (int k = 0; k < K; ++k, ++c) {
                const double *pc = p->psf + 6 * k;
                const double n_ = nu[8 * i + j];
                const double s11 = pc[3] + n_ * x11, s12 = pc[4] + n_ * x12, s22 = pc[5] + n_ * x22;
                const double det = s11 * s22 - s12 * s12, idet = 1.0 / det;
                const double z = (pc[0] * eta[8 * i + j]) / (sqrt(det) * (2.0 * M_PI));
                RComp *o = &comps[c];
                o->p11 = s22 * idet; o->p12 = -s12 * idet; o->p22 = s11 * idet;
                o->xi1 = pc[1]; o->xi2 = pc[2];
                o->w0 = z * (i == 0 ? vs[2] : 1.0 - vs[2]);
                o->wd = i == 0 ? z : -z;
                o->nu = n_;
            }
    double El0, Ell0, El1, Ell1;
    r_brightness(vs, 0, band0, &El0, &Ell0);
    r_brightness(vs, 1, band0, &El1, &Ell1);
    si->c0 = vs[26] * El0; si->c1 = vs[27] * El1; si->q0 = vs[26] * Ell0; si->q1 = vs[27] * Ell1;
}

static inline void bw(double f, double w[4]) {
    const double o = 1.0 - f;
    w[0] = o * o * o / 6; w[1] = 2.0 / 3 - f * f + f * f * f * 0.5; w[2] = 2.0 / 3 - o * o + o * o * o * 0.5; w[3] = f * f * f / 6;
}
static inline void bdw(double f, double dw[4], double ddw[4]) {
    const double o = 1.0 - f;
    dw[0] = -0.5 * o * o; dw[1] = -2 * f + 1.5 * f * f; dw[2] = 2 * o - 1.5 * o * o; dw[3] = 0.5 * f * f;
    ddw[0] = o; ddw[1] = -2 + 3 * f; ddw[2] = -2 + 3 * o; ddw[3] = f;
}

/* star density: value (and derivatives with respect to the position when g != NULL: g[2], h[3] = 11, 12, 22) */
static double r_star(const double *coef, double xh, double xw, double *g, double *h) {
    int ix = (int)floor(xh); ix = ix < 1 ? 1 : (ix > 50 ? 50 : ix);
    int iy = (int)floor(xw); iy = iy < 1 ? 1 : (iy > 50 ? 50 : iy);
    const double fx = xh - ix, fy = xw - iy;
    double wx[4], wy[4], dwx[4], ddwx[4], dwy[4], ddwy[4];
    bw(fx, wx); bw(fy, wy);
    if (g) { bdw(fx, dwx, ddwx); bdw(fy, dwy, ddwy); }
    const double *cc = coef + (ix - 1) + COEF * (iy - 1);
    double y = 0, yx = 0, yy = 0, yxx = 0, yxy = 0, yyy = 0;
    for (int b = 0; b < 4; ++b) {
        const double *cb = cc + COEF * b;
        const double r = cb[0] * wx[0] + cb[1] * wx[1] + cb[2] * wx[2] + cb[3] * wx[3];
        y += r * wy[b];
        if (g) {
            const double rx = cb[0] * dwx[0] + cb[1] * dwx[1] + cb[2] * dwx[2] + cb[3] * dwx[3];
            const double rxx = cb[0] * ddwx[0] + cb[1] * ddwx[1] + cb[2] * ddwx[2] + cb[3] * ddwx[3];
            yx += rx * wy[b]; yxx += rxx * wy[b]; yy += r * dwy[b]; yxy += rx * dwy[b]; yyy += r * ddwy[b];
        }
    }
    double gv, gp, gpp;
    if (y < 0) { gv = 1e-3 * exp(y); gp = gv; gpp = gv; } else { gv = 1e-3 * (y + 1.0); gp = 1e-3; gpp = 0.0; }
    if (g) {
        const double ym1 = -yx, ym2 = -yy; /* d(index)/dm = -1 */
        g[0] = gp * ym1; g[1] = gp * ym2;
        h[0] = gpp * ym1 * ym1 + gp * yxx; h[1] = gpp * ym1 * ym2 + gp * yxy; h[2] = gpp * ym2 * ym2 + gp * yyy;
    }
    return gv;
}

static double r_galaxy_value(const RComp *tc, int NC, double dx, double dy) {
    double v = 0;
    for (int c = 0; c < NC; ++c) {
        const RComp *k = &tc[c];
        const double d1 = dx - k->xi1, d2 = dy - k->xi2;
        const double u = k->p11 * d1 + k->p12 * d2, vv = k->p12 * d1 + k->p22 * d2;
        v += k->w0 * exp(-0.5 * (d1 * u + d2 * vv));
    }
    return v;
}

/* the 24 component sums: Gaussian derivatives with respect to the mean are Hermite polynomials in (u, v) = P d,
 * d/dSigma = (1/2) d2/dx2, Sigma = tau + nu Xi */
typedef struct {
    double S0, S0d, S1x, S1y, S1xd, S1yd, S2a, S2b, S2c, S2an, S2bn, S2cn, S2ad, S2bd, S2cd, S3a, S3b, S3c, S3d, S4a, S4b, S4c, S4d, S4e;
} GSums;

static void r_galaxy_sums(const RComp *tc, int NC, double dx, double dy, GSums *G) {
    memset(G, 0, sizeof *G);
    for (int c = 0; c < NC; ++c) {
        const RComp *k = &tc[c];
        const double d1 = dx - k->xi1, d2 = dy - k->xi2;
        const double u = k->p11 * d1 + k->p12 * d2, v = k->p12 * d1 + k->p22 * d2;
        const double e = exp(-0.5 * (d1 * u + d2 * v));
        const double f = k->w0 * e, fd = k->wd * e, fn = f * k->nu, fdn = fd * k->nu, fnn = fn * k->nu;
        const double ha = u * u - k->p11, hb = u * v - k->p12, hc = v * v - k->p22;
        G->S0 += f; G->S0d += fd; G->S1x += u * f; G->S1y += v * f; G->S1xd += u * fd; G->S1yd += v * fd;
        G->S2a += ha * f; G->S2b += hb * f; G->S2c += hc * f;
        G->S2an += ha * fn; G->S2bn += hb * fn; G->S2cn += hc * fn;
        G->S2ad += ha * fdn; G->S2bd += hb * fdn; G->S2cd += hc * fdn;
        const double h3a = u * (ha - 2 * k->p11), h3b = v * ha - 2 * u * k->p12, h3c = u * hc - 2 * v * k->p12,
                     h3d = v * (hc - 2 * k->p22);
        G->S3a += h3a * fn; G->S3b += h3b * fn; G->S3c += h3c * fn; G->S3d += h3d * fn;
        const double h4a = u * h3a - 3 * ha * k->p11, h4b = v * h3a - 3 * ha * k->p12,
                     h4c = u * h3c - 2 * hb * k->p12 - hc * k->p11, h4d = u * h3d - 3 * hc * k->p12,
                     h4e = v * h3d - 3 * hc * k->p22;
        G->S4a += h4a * fnn; G->S4b += h4b * fnn; G->S4c += h4c * fnn; G->S4d += h4d * fnn; G->S4e += h4e * fnn;
    }
}

/* one target: value, gradient (44), Hessian (44 x 44 col-major), counters */
static int reduced_elbo_one(const celeste_problem_t *pr, const double *coefs_all, const double *eta, const double *nu,
                            const double *vp, int t, uint32_t flags, double *v_out, double *d_out, double *h_out,
                            int64_t *cnt) {
    const int N = pr->n_images, K = pr->psf_K, NC = 14 * K;
    const int want_h = (flags & CELESTE_FLAG_HESS) != 0, want_g = want_h || (flags & CELESTE_FLAG_GRAD);
    const double *vs = vp + (size_t)t * P;
    const int64_t nb0 = pr->nbr_offsets ? pr->nbr_offsets[t] : 0, nb1 = pr->nbr_offsets ? pr->nbr_offsets[t + 1] : 0;
    const int nnb = (int)(nb1 - nb0);
    for (int k = 0; k < P; ++k) if (!isfinite(vs[k])) return CELESTE_ERR_NONFINITE_INPUT;
    for (int q = 0; q < nnb; ++q)
        for (int k = 0; k < P; ++k) if (!isfinite(vp[(size_t)pr->nbr_index[nb0 + q] * P + k])) return CELESTE_ERR_NONFINITE_INPUT;

    double value = 0, d[NP], h[NP * NP];
    memset(d, 0, sizeof d); memset(h, 0, sizeof h);
    int64_t n_act = 0, n_inact = 0;
    RComp *tc = (RComp *)malloc(sizeof(RComp) * NC * (1 + (size_t)nnb));
    RSrc *ns = (RSrc *)malloc(sizeof(RSrc) * (1 + (size_t)nnb));

    /* shape Jacobian / second-derivative tensor of Xi with respect to (axis ratio, angle, radius) */
    double jsh[9], tsh[27];
    {
        double x11, x12, x22;
        r_bvn_cov(vs[3], vs[4], vs[5], &x11, &x12, &x22);
        const double ab = vs[3], ang = vs[4], r = vs[5], sn = sin(ang), cs = cos(ang);
        const double cos_sin = cs * sn, sin_sq = sn * sn, cos_sq = cs * cs, r2 = r * r;
        const double c1 = 2 * ab * r2, c2 = r2 * (ab * ab - 1), a = 2 * r2, b2 = a * (ab * ab - 1);
        jsh[0] = c1 * sin_sq; jsh[1] = -c1 * cos_sin; jsh[2] = c1 * cos_sq;
        jsh[3] = c2 * (2 * cos_sin); jsh[4] = c2 * (sin_sq - cos_sq); jsh[5] = c2 * (-2 * cos_sin);
        jsh[6] = 2 * x11 / r; jsh[7] = 2 * x12 / r; jsh[8] = 2 * x22 / r;
        tsh[0] = sin_sq * a; tsh[1] = -cos_sin * a; tsh[2] = cos_sq * a;
        tsh[3] = 2 * cos_sin * a * ab; tsh[4] = (sin_sq - cos_sq) * a * ab; tsh[5] = -2 * cos_sin * a * ab;
        tsh[6] = 2 * jsh[0] / r; tsh[7] = 2 * jsh[1] / r; tsh[8] = 2 * jsh[2] / r;
        tsh[9] = tsh[3]; tsh[10] = tsh[4]; tsh[11] = tsh[5];
        tsh[12] = (cos_sq - sin_sq) * b2; tsh[13] = 2 * cos_sin * b2; tsh[14] = (sin_sq - cos_sq) * b2;
        tsh[15] = 2 * jsh[3] / r; tsh[16] = 2 * jsh[4] / r; tsh[17] = 2 * jsh[5] / r;
        tsh[18] = tsh[6]; tsh[19] = tsh[7]; tsh[20] = tsh[8];
        tsh[21] = tsh[15]; tsh[22] = tsh[16]; tsh[23] = tsh[17];
        tsh[24] = 2 * x11 / r2; tsh[25] = 2 * x12 / r2; tsh[26] = 2 * x22 / r2;
    }

    for (int n = 0; n < N; ++n) {
        const celeste_patch_t *pa = oracle_patch_at(pr, t, n);
        if (pa->H2 <= 0 || pa->W2 <= 0) continue;
        const celeste_image_t *img = &pr->images[n];
        const int b = img->band - 1;
        r_prep(pa, K, vs, b, eta, nu, tc, &ns[0]);
        for (int q = 0; q < nnb; ++q) {
            const int s2 = pr->nbr_index[nb0 + q];
            r_prep(oracle_patch_at(pr, s2, n), K, vp + (size_t)s2 * P, b, eta, nu, tc + (size_t)NC * (1 + q), &ns[1 + q]);
        }
        const double *tcoef = coefs_all + (size_t)pa->stamp * COEF * COEF;
        const RSrc si = ns[0];
        const double c0 = si.c0, c1 = si.c1, q0 = si.q0, q1 = si.q1;
        double rec[NREC];
        memset(rec, 0, sizeof rec);
        for (int w2 = 1; w2 <= pa->W2; ++w2) for (int h2 = 1; h2 <= pa->H2; ++h2) {
            const int hh = pa->off_h + h2, ww = pa->off_w + w2;
            const size_t gi = (size_t)(hh - 1) + (size_t)img->H * (ww - 1);
            const float xf = img->pixels[gi];
            if (isnan(xf)) continue;
            if (pa->bitmap && !pa->bitmap[(h2 - 1) + (size_t)pa->H2 * (w2 - 1)]) continue;
            double Ebar = (double)img->sky[gi], Vbar = 0;
            for (int q = 0; q < nnb; ++q) {
                const int s2 = pr->nbr_index[nb0 + q];
                const celeste_patch_t *Q = oracle_patch_at(pr, s2, n);
                const int ph2 = hh - Q->off_h, pw2 = ww - Q->off_w;
                if (!(ph2 >= 1 && ph2 <= Q->H2 && pw2 >= 1 && pw2 < Q->W2)) continue;
                if (Q->bitmap) { if (!Q->bitmap[(ph2 - 1) + (size_t)Q->H2 * (pw2 - 1)]) continue; }
                else if (isnan(img->pixels[gi])) continue;
                const RSrc *sq = &ns[1 + q];
                const double f0 = r_star(coefs_all + (size_t)Q->stamp * COEF * COEF, hh + (26.0 - sq->m1), ww + (26.0 - sq->m2), NULL, NULL);
                const double f1 = r_galaxy_value(tc + (size_t)NC * (1 + q), NC, hh - sq->m1, ww - sq->m2);
                const double A = sq->c0 * f0 + sq->c1 * f1;
                Ebar += A; Vbar += (sq->q0 * f0 * f0 + sq->q1 * f1 * f1) - A * A;
                n_inact += 1;
            }
            const int own = w2 < pa->W2; /* 1 <= w2 < W2, elbo_objective.jl:349 */
            const double x = (double)xf, iota = (double)img->nelec_per_nmgy[hh - 1];
            const double log_iota = (double)(float)log((double)img->nelec_per_nmgy[hh - 1]);
            double f0 = 0, f1 = 0, sg[2] = {0, 0}, sh[3] = {0, 0, 0};
            GSums G; memset(&G, 0, sizeof G);
            if (own) {
                n_act += 1;
                if (want_g) {
                    r_galaxy_sums(tc, NC, hh - si.m1, ww - si.m2, &G);
                    f1 = G.S0;
                    f0 = r_star(tcoef, hh + (26.0 - si.m1), ww + (26.0 - si.m2), sg, sh);
                } else {
                    f1 = r_galaxy_value(tc, NC, hh - si.m1, ww - si.m2);
                    f0 = r_star(tcoef, hh + (26.0 - si.m1), ww + (26.0 - si.m2), NULL, NULL);
                }
            }
            const double A = c0 * f0 + c1 * f1, B = q0 * f0 * f0 + q1 * f1 * f1;
            const double E = Ebar + A, V = Vbar + (B - A * A);
            const double iE = 1.0 / E, iE2 = iE * iE, iE3 = iE2 * iE;
            rec[0] += x * (log_iota + (log(E) - V * (0.5 * iE2))) - iota * E - lgamma(x + 1.0);
            if (!own || !want_g) continue;
            const double w1 = x * (iE + V * iE3) - iota, w2_ = -0.5 * x * iE2, w11 = -x * (iE2 + 3.0 * V * iE2 * iE2), w12 = x * iE3;
            const double alpha = w1 - 2.0 * A * w2_, beta = w11 - 2.0 * w2_ - 4.0 * A * w12;
            /* gradients of f1 (gg), A (dA) and B (dB) with respect to (m1, m2, dev, Xi11, Xi12, Xi22) */
            const double gg[6] = {G.S1x, G.S1y, G.S0d, 0.5 * G.S2an, G.S2bn, 0.5 * G.S2cn};
            double dA[6], dB[6];
            for (int k = 0; k < 6; ++k) {
                const double s_ = k < 2 ? sg[k] : 0.0;
                dA[k] = c0 * s_ + c1 * gg[k];
                dB[k] = 2.0 * q0 * f0 * s_ + 2.0 * q1 * f1 * gg[k];
            }
            rec[1] += alpha * f0; rec[2] += alpha * f1; rec[3] += w2_ * f0 * f0; rec[4] += w2_ * f1 * f1;
            for (int k = 0; k < 6; ++k) rec[5 + k] += alpha * dA[k] + w2_ * dB[k];
            if (!want_h) continue;
            const double k1 = alpha * c1 + 2.0 * w2_ * q1 * f1, k0 = alpha * c0 + 2.0 * w2_ * q0 * f0;
            const double r1 = 2.0 * w2_ * q1, r0 = 2.0 * w2_ * q0;
            /* second derivatives of f1, upper triangle (g <= g2) */
            double gh[6][6];
            memset(gh, 0, sizeof gh);
            gh[0][0] = G.S2a; gh[0][1] = G.S2b; gh[0][2] = G.S1xd; gh[0][3] = 0.5 * G.S3a; gh[0][4] = G.S3b; gh[0][5] = 0.5 * G.S3c;
            gh[1][1] = G.S2c; gh[1][2] = G.S1yd; gh[1][3] = 0.5 * G.S3b; gh[1][4] = G.S3c; gh[1][5] = 0.5 * G.S3d;
            gh[2][3] = 0.5 * G.S2ad; gh[2][4] = G.S2bd; gh[2][5] = 0.5 * G.S2cd;
            gh[3][3] = 0.25 * G.S4a; gh[3][4] = 0.5 * G.S4b; gh[3][5] = 0.25 * G.S4c;
            gh[4][4] = G.S4c; gh[4][5] = 0.5 * G.S4d; gh[5][5] = 0.25 * G.S4e;
            const double fv[2] = {f0, f1};
            for (int i = 0; i < ZV; ++i) for (int j = i; j < ZV; ++j) {
                double e;
                if (j < 2) e = beta * fv[i] * fv[j];                                    /* (c, c) */
                else if (i < 2 && j < 4) e = w12 * fv[i] * (fv[j - 2] * fv[j - 2]);      /* (c, q) */
                else if (j < 4) e = 0.0;                                                 /* (q, q) */
                else if (i < 4) {
                    const int g2 = j - 4, star = (i & 1) == 0;
                    const double fi = star ? f0 : f1;
                    const double fig = star ? (g2 < 2 ? sg[g2] : 0.0) : gg[g2];
                    if (i < 2) e = alpha * fig + fi * (beta * dA[g2] + w12 * dB[g2]);      /* (c, geo) */
                    else e = 2.0 * w2_ * fi * fig + w12 * (fi * fi) * dA[g2];             /* (q, geo) */
                } else {                                                                 /* (geo, geo) */
                    const int g = i - 4, g2 = j - 4;
                    e = k1 * gh[g][g2] + r1 * gg[g] * gg[g2] + beta * dA[g] * dA[g2] + w12 * (dA[g] * dB[g2] + dB[g] * dA[g2]);
                    if (g2 < 2) e += k0 * sh[g + g2] + r0 * sg[g] * sg[g2];
                }
                rec[hidx(i, j)] += e;
            }
        }
        value += rec[0];
        if (!want_g) continue;
        /* chain rule to the 28 parameters with likelihood derivatives: z = (c0, c1, q0, q1, m1, m2, dev, Xi) */
        double J[ZV][NP], El[2], Ell[2], kap[2][10], lam[2][10];
        memset(J, 0, sizeof J);
        for (int ty = 0; ty < 2; ++ty) {
            r_brightness(vs, ty, b, &El[ty], &Ell[ty]);
            for (int q = 0; q < 10; ++q) {
                double ka = 0, la = 0;
                if (q == 0) { ka = 1; la = 2; }
                else if (q == 1) { ka = .5; la = 2; }
                else {
                    const int c = (q - 2) & 3, is_var = q >= 6;
                    int on; double sgn;
                    if (c == 2) { on = b >= 3; sgn = 1; } else if (c == 3) { on = b >= 4; sgn = 1; }
                    else if (c == 1) { on = b <= 1; sgn = -1; } else { on = b <= 0; sgn = -1; }
                    if (on) { ka = is_var ? .5 : sgn; la = is_var ? 2 : 2 * sgn; }
                }
                kap[ty][q] = ka; lam[ty][q] = la;
            }
        }
        int slot[NP], ptype[NP]; /* brightness slot (bids order, -1: is_star) and type of every brightness parameter */
        for (int p = 0; p < NP; ++p) {
            slot[p] = -2; ptype[p] = -1;
            if (p >= 6) {
                ptype[p] = p < 10 ? ((p - 6) & 1) : (p < 26 ? (((p - 10) >> 2) & 1) : p - 26);
                slot[p] = p < 8 ? 0 : (p < 10 ? 1 : (p < 18 ? 2 + ((p - 10) & 3) : (p < 26 ? 6 + ((p - 18) & 3) : -1)));
            }
        }
        for (int p = 0; p < 2; ++p) { J[4][p] = pa->wcs_jacobian[0 + 2 * p]; J[5][p] = pa->wcs_jacobian[1 + 2 * p]; }
        J[6][2] = 1.0;
        for (int p = 3; p < 6; ++p) for (int r = 7; r < 10; ++r) J[r][p] = jsh[(r - 7) + 3 * (p - 3)];
        for (int p = 6; p < NP; ++p) {
            const int ty = ptype[p];
            J[ty][p] = slot[p] < 0 ? El[ty] : vs[26 + ty] * El[ty] * kap[ty][slot[p]];
            J[2 + ty][p] = slot[p] < 0 ? Ell[ty] : vs[26 + ty] * Ell[ty] * lam[ty][slot[p]];
        }
        for (int p = 0; p < NP; ++p) { double s = 0; for (int r = 0; r < ZV; ++r) s += J[r][p] * rec[1 + r]; d[p] += s; }
        if (!want_h) continue;
        double HJ[ZV][NP]; /* H_z J */
        for (int r1_ = 0; r1_ < ZV; ++r1_) for (int p = 0; p < NP; ++p) {
            double s = 0;
            for (int r2 = 0; r2 < ZV; ++r2) s += rec[hidx(r1_ < r2 ? r1_ : r2, r1_ < r2 ? r2 : r1_)] * J[r2][p];
            HJ[r1_][p] = s;
        }
        for (int p1 = 0; p1 < NP; ++p1) for (int p2 = p1; p2 < NP; ++p2) {
            double s = 0;
            for (int r = 0; r < ZV; ++r) s += J[r][p1] * HJ[r][p2];
            if (p1 >= 3 && p1 < 6 && p2 >= 3 && p2 < 6)
                for (int g = 0; g < 3; ++g) s += rec[1 + 7 + g] * tsh[g + 3 * (p1 - 3) + 9 * (p2 - 3)];
            else if (p1 >= 6 && ptype[p1] == ptype[p2]) {
                const int ty = ptype[p1], b1 = slot[p1], b2 = slot[p2];
                const double gc = rec[1 + ty], gq = rec[1 + 2 + ty], ai = vs[26 + ty];
                if (b1 >= 0 && b2 >= 0) s += ai * (gc * El[ty] * kap[ty][b1] * kap[ty][b2] + gq * Ell[ty] * lam[ty][b1] * lam[ty][b2]);
                else if (b1 >= 0) s += gc * El[ty] * kap[ty][b1] + gq * Ell[ty] * lam[ty][b1];
                else if (b2 >= 0) s += gc * El[ty] * kap[ty][b2] + gq * Ell[ty] * lam[ty][b2];
            }
            h[p1 + NP * p2] += s;
        }
    }
    free(tc); free(ns);

    double kv = 0, kd[P], *kh = NULL;
    memset(kd, 0, sizeof kd);
    if (flags & CELESTE_FLAG_KL) {
        kh = (double *)malloc(sizeof(double) * P * P);
        celeste_oracle_subtract_kl(pr->prior, vs, &kv, kd, kh);
    }
    int status = CELESTE_OK;
    value += kv;
    if (!isfinite(value)) status = CELESTE_ERR_NONFINITE_RESULT;
    if (v_out) *v_out = value;
    if (want_g && d_out) for (int p = 0; p < P; ++p) {
        const double x = (p < NP ? d[p] : 0.0) + kd[p];
        if (!isfinite(x)) status = CELESTE_ERR_NONFINITE_RESULT;
        d_out[p] = x;
    }
    if (want_h && h_out) for (int p2 = 0; p2 < P; ++p2) for (int p1 = 0; p1 < P; ++p1) {
        const int a = p1 < p2 ? p1 : p2, c = p1 < p2 ? p2 : p1;
        double x = (c < NP) ? h[a + NP * c] : 0.0;
        if (kh) x += kh[a + P * c];
        if (!isfinite(x)) status = CELESTE_ERR_NONFINITE_RESULT;
        h_out[p1 + P * p2] = x;
    }
    free(kh);
    if (cnt) { cnt[0] = n_act; cnt[1] = n_inact; }
    return status;
}

int celeste_reduced_elbo_batch(const celeste_problem_t *pr, const double *vp, int32_t n_targets, const int32_t *targets,
                               uint32_t flags, double *v, double *d, double *h, int64_t *counters, int32_t *status,
                               int32_t n_threads) {
    if (!pr || !vp || !targets) return CELESTE_ERR_INVALID_ARG;
    double eta[16], nu[16];
    celeste_oracle_galaxy_prototypes(eta, nu);
    double *coefs = (double *)malloc(sizeof(double) * (size_t)(pr->n_stamps > 0 ? pr->n_stamps : 1) * COEF * COEF);
    for (int k = 0; k < pr->n_stamps; ++k) celeste_oracle_spline_coefs(pr->stamps + (size_t)k * 51 * 51, coefs + (size_t)k * COEF * COEF);
    int worst = CELESTE_OK;
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads > 0 ? n_threads : 1)
#endif
    for (int q = 0; q < n_targets; ++q) {
        int64_t cnt[2] = {0, 0};
        int st = (targets[q] < 0 || targets[q] >= pr->n_sources) ? CELESTE_ERR_INVALID_ARG
                 : reduced_elbo_one(pr, coefs, eta, nu, vp, targets[q], flags, v ? v + q : NULL, d ? d + (size_t)q * P : NULL,
                                    h ? h + (size_t)q * P * P : NULL, cnt);
        if (counters) { counters[2 * q] = cnt[0]; counters[2 * q + 1] = cnt[1]; }
        if (status) status[q] = st;
        if (st != CELESTE_OK) {
#ifdef _OPENMP
#pragma omp critical
#endif
            worst = st;
        }
    }
    free(coefs);
    return worst;
}
