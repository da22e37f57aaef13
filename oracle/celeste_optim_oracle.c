/*
 * celeste_optim_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the caller of the ELBO hot path:
 *   ConstraintTransforms (src/deterministic_vi/ConstraintTransforms.jl): to_free!/to_bound!/enforce!,
 *     propagate_derivatives! (analytic Jacobian / Hessian instead of nested ForwardDiff),
 *   ElboMaximize.maximize! (src/deterministic_vi/ElboMaximize.jl:63-108, 161-242) driving a Newton
 *     trust-region iteration.
 * The trust-region method itself lives in the third-party package Optim.jl (REQUIRE: "Optim 0.7.4+", not
 * vendored, unpinned): it is restated here from its published algorithm -- Nocedal & Wright, Numerical
 * Optimization, Alg. 4.1 for the radius update (eta = 0.1, shrink below rho = 0.25 by 1/4, grow above 0.75 by 2
 * when the step is not interior), and Optim's solve_tr_subproblem! rules for the sub-problem (see
 * celeste_oracle_solve_tr below), solved in the eigenbasis of the Hessian (cyclic Jacobi).  PARITY UNPINNED for
 * iterate-by-iterate agreement with Optim.jl; the reference's own optimiser tests only assert recovery tolerances
 * (test/test_optimization.jl:10-32), which tests/ mirrors.
 *
 * The ELBO evaluations come from celeste_oracle.c (celeste_oracle_elbo).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/celeste_mi355x.h"

#define P 44
#define NF 41

int celeste_oracle_elbo(const celeste_problem_t *pr, const double *vp, int32_t target, uint32_t flags,
                        double *v, double *d, double *h, int64_t *n_active_px, int64_t *n_inactive_px);

/* ---- constraints (ElboMaximize.elbo_constraints, ElboMaximize.jl:63-93) -------------------------------- */
typedef struct { double lo[26], hi[26], scale[26]; } Boxes;              /* bound index == free index, 0..25 */
static const double SIMPLEX_LO[3] = {0.005, 0.01 / 8, 0.01 / 8};
static const int SIMPLEX_N[3] = {2, 8, 8};
static const int SIMPLEX_BOUND0[3] = {26, 28, 36};
static const int SIMPLEX_FREE0[3] = {26, 27, 34};

static void make_boxes(Boxes *b, const double *vs, double loc_width, double loc_scale) {
    for (int i = 0; i < 26; ++i) b->scale[i] = 1.0;
    b->lo[0] = vs[0] - loc_width; b->hi[0] = vs[0] + loc_width; b->scale[0] = loc_scale;
    b->lo[1] = vs[1] - loc_width; b->hi[1] = vs[1] + loc_width; b->scale[1] = loc_scale;
    b->lo[2] = 1e-2; b->hi[2] = 0.99;          /* gal_frac_dev */
    b->lo[3] = 1e-2; b->hi[3] = 0.99;          /* gal_axis_ratio */
    b->lo[4] = -10.0; b->hi[4] = 10.0;         /* gal_angle */
    b->lo[5] = 0.10; b->hi[5] = 70.0;          /* gal_radius_px */
    for (int i = 6; i < 8; ++i) { b->lo[i] = -1.0; b->hi[i] = 10.0; }     /* flux_loc */
    for (int i = 8; i < 10; ++i) { b->lo[i] = 1e-4; b->hi[i] = 0.10; }    /* flux_scale */
    for (int i = 10; i < 18; ++i) { b->lo[i] = -10.0; b->hi[i] = 10.0; }  /* color_mean */
    for (int i = 18; i < 26; ++i) { b->lo[i] = 1e-4; b->hi[i] = 1.0; }    /* color_var */
}

/* enforce! (ConstraintTransforms.jl:225-253) */
static void enforce(double *vs, const Boxes *b) {
    for (int i = 0; i < 26; ++i)
        if (!(b->lo[i] < vs[i] && vs[i] < b->hi[i]))
            vs[i] = fmax(fmin(vs[i], nextafter(b->hi[i], -INFINITY)), nextafter(b->lo[i], INFINITY));
    for (int g = 0; g < 3; ++g) {
        const int n = SIMPLEX_N[g]; double *x = vs + SIMPLEX_BOUND0[g]; const double lo = SIMPLEX_LO[g];
        double sum = 0;
        for (int i = 0; i < n; ++i) {
            if (!(lo < x[i] && x[i] < 1.0)) x[i] = fmax(fmin(x[i], nextafter(1.0, -INFINITY)), nextafter(lo, INFINITY));
            sum += x[i];
        }
        /* isapprox(sum, 1): |sum - 1| <= sqrt(eps) * max(|sum|, 1) */
        if (!(fabs(sum - 1.0) <= 1.4901161193847656e-08 * fmax(fabs(sum), 1.0))) {
            const double rescale = (1 - n * lo) / (sum - n * lo);
            for (int i = 0; i < n; ++i) x[i] = nextafter(lo, INFINITY) + rescale * (x[i] - lo);
        }
    }
}

/* to_free! (ConstraintTransforms.jl:84-126) */
void celeste_oracle_to_free(const double *vs, const Boxes *b, double *x) {
    for (int i = 0; i < 26; ++i) {
        const double u = (vs[i] - b->lo[i]) / (b->hi[i] - b->lo[i]);
        x[i] = -log(1.0 / u - 1) * b->scale[i];
    }
    for (int g = 0; g < 3; ++g) {
        const int n = SIMPLEX_N[g]; const double *bd = vs + SIMPLEX_BOUND0[g]; const double lo = SIMPLEX_LO[g];
        const double log_last = log((bd[n - 1] - lo) / (1 - n * lo));
        for (int i = 0; i < n - 1; ++i) x[SIMPLEX_FREE0[g] + i] = 1.0 * (log((bd[i] - lo) / (1 - n * lo)) - log_last);
    }
}

/* to_bound! with first and second derivatives.  J[a + 44 i] = d bound_a / d free_i;
 * for the Hessian of bound_a only its own group's free indices matter: returned through a callback-free
 * dense tensor T2[a][i][j] would be 44*41*41; instead the contraction sum_a dvec[a] * d2 bound_a/dfree_i dfree_j
 * is accumulated directly into H2 (41 x 41) when dvec != NULL. */
void celeste_oracle_to_bound(const double *x, const Boxes *b, double *vs, double *J, const double *dvec, double *H2) {
    if (J) memset(J, 0, sizeof(double) * P * NF);
    if (H2) memset(H2, 0, sizeof(double) * NF * NF);
    for (int i = 0; i < 26; ++i) {
        const double s = 1.0 / (1.0 + exp(-x[i] / b->scale[i]));
        const double w = b->hi[i] - b->lo[i];
        vs[i] = s * w + b->lo[i];
        if (J) J[i + P * i] = w * s * (1 - s) / b->scale[i];
        if (H2 && dvec) H2[i + NF * i] += dvec[i] * w * s * (1 - s) * (1 - 2 * s) / (b->scale[i] * b->scale[i]);
    }
    for (int g = 0; g < 3; ++g) {
        const int n = SIMPLEX_N[g], b0 = SIMPLEX_BOUND0[g], f0 = SIMPLEX_FREE0[g];
        const double lo = SIMPLEX_LO[g], sc = 1 - n * lo;
        double z[8], m = x[f0];
        for (int i = 0; i < n - 1; ++i) { z[i] = x[f0 + i]; if (z[i] > m) m = z[i]; }
        const double exp_neg_m = exp(-m);
        double sum = exp_neg_m, p[8];
        for (int i = 0; i < n - 1; ++i) { p[i] = exp(z[i] - m); sum += p[i]; }
        for (int i = 0; i < n - 1; ++i) p[i] = p[i] / sum;
        p[n - 1] = (1.0 / sum) * exp_neg_m;
        for (int i = 0; i < n; ++i) vs[b0 + i] = sc * p[i] + lo;
        /* softmax derivatives with the last logit fixed at 0: dp_a/dx_j = p_a (delta_aj - p_j) */
        for (int a = 0; a < n; ++a) for (int j = 0; j < n - 1; ++j) {
            const double dj = p[a] * ((a == j) - p[j]);
            if (J) J[(b0 + a) + P * (f0 + j)] = sc * dj;
            if (H2 && dvec) for (int k = 0; k < n - 1; ++k) {
                const double d2 = p[a] * (((a == j) - p[j]) * ((a == k) - p[k]) - p[j] * ((j == k) - p[k]));
                H2[(f0 + j) + NF * (f0 + k)] += dvec[b0 + a] * sc * d2;
            }
        }
    }
}

/* propagate_derivatives! (ConstraintTransforms.jl:373-457): free gradient J' d, free Hessian J' h J + sum_a d_a H_a */
static void propagate(const double *x, const Boxes *b, const double *d, const double *h, double *gf, double *Hf) {
    double vs[P]; double *J = (double *)malloc(sizeof(double) * P * NF), *W = (double *)malloc(sizeof(double) * P * NF);
    celeste_oracle_to_bound(x, b, vs, J, d, Hf);
    for (int i = 0; i < NF; ++i) { double s = 0; for (int a = 0; a < P; ++a) s += J[a + P * i] * d[a]; gf[i] = s; }
    for (int i = 0; i < NF; ++i) for (int a = 0; a < P; ++a) { double s = 0; for (int c = 0; c < P; ++c) s += h[a + P * c] * J[c + P * i]; W[a + P * i] = s; }
    for (int i = 0; i < NF; ++i) for (int j = 0; j < NF; ++j) { double s = 0; for (int a = 0; a < P; ++a) s += J[a + P * i] * W[a + P * j]; Hf[i + NF * j] += s; }
    for (int i = 0; i < NF; ++i) for (int j = 0; j < i; ++j) { double s = 0.5 * (Hf[i + NF * j] + Hf[j + NF * i]); Hf[i + NF * j] = s; Hf[j + NF * i] = s; }
    free(J); free(W);
}
void celeste_oracle_propagate(const double *x, const double *vs0, double loc_width, double loc_scale, const double *d,
                              const double *h, double *gf, double *Hf) {
    Boxes b; make_boxes(&b, vs0, loc_width, loc_scale); propagate(x, &b, d, h, gf, Hf);
}

/* ---- symmetric eigen-decomposition: cyclic Jacobi --------------------------------------------------------- */
/* A (n x n, column-major, symmetric) -> eigenvalues w (ascending) and eigenvectors V (columns) */
void celeste_oracle_jacobi_eig(int n, const double *A_in, double *w, double *V) {
    double *A = (double *)malloc(sizeof(double) * n * n);
    memcpy(A, A_in, sizeof(double) * n * n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) V[i + n * j] = (i == j);
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0, diag = 0;
        for (int p = 0; p < n; ++p) { diag += A[p + n * p] * A[p + n * p]; for (int q = p + 1; q < n; ++q) off += A[p + n * q] * A[p + n * q]; }
        if (off <= 1e-30 * (diag + off) || off == 0) break;
        for (int p = 0; p < n - 1; ++p) for (int q = p + 1; q < n; ++q) {
            const double apq = A[p + n * q];
            if (apq == 0) continue;
            const double theta = (A[q + n * q] - A[p + n * p]) / (2 * apq);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
            const double c = 1 / sqrt(t * t + 1), s = t * c;
            for (int k = 0; k < n; ++k) { /* rows p, q */
                const double akp = A[p + n * k], akq = A[q + n * k];
                A[p + n * k] = c * akp - s * akq; A[q + n * k] = s * akp + c * akq;
            }
            for (int k = 0; k < n; ++k) { /* columns p, q */
                const double akp = A[k + n * p], akq = A[k + n * q];
                A[k + n * p] = c * akp - s * akq; A[k + n * q] = s * akp + c * akq;
                const double vkp = V[k + n * p], vkq = V[k + n * q];
                V[k + n * p] = c * vkp - s * vkq; V[k + n * q] = s * vkp + c * vkq;
            }
        }
    }
    for (int i = 0; i < n; ++i) w[i] = A[i + n * i];
    /* selection sort ascending, permuting eigenvectors */
    for (int i = 0; i < n - 1; ++i) {
        int m = i; for (int j = i + 1; j < n; ++j) if (w[j] < w[m]) m = j;
        if (m != i) {
            double t = w[i]; w[i] = w[m]; w[m] = t;
            for (int k = 0; k < n; ++k) { double u = V[k + n * i]; V[k + n * i] = V[k + n * m]; V[k + n * m] = u; }
        }
    }
    free(A);
}

/* ---- trust-region sub-problem: min g's + 0.5 s'Hs, |s| <= delta.
 * Optim.jl (third-party, not vendored under /root/reference; Celeste's REQUIRE asks for Optim >= 0.7.4) solves it
 * in solve_tr_subproblem! of its NewtonTrustRegion method.  Restated from that published routine (N&W section
 * 4.3, Alg. 4.3): eigen-decomposition; the unconstrained Newton step when the smallest eigenvalue is >= 1e-8 and
 * the step fits; otherwise lambda starts at lambda_lb = -w_min + max(1e-8, 1e-8 (w_max - w_min)); the hard case
 * only when w_min < 0 and g is orthogonal (1e-10) to every eigenvector whose eigenvalue is within 1e-10 of w_min;
 * else Newton iterations on lambda (tolerance 1e-10, halving towards lambda_lb when the update undershoots it),
 * the step being the one of the last factorisation.  Optim stops after max_iters = 5 such iterations whether or
 * not they converged; here they run to convergence (<= 20; the iterates increase monotonically from the left of
 * the root, so a non-increasing update means the rounding floor is reached).  parity unpinned: Optim's source is
 * not available here; pinned only through the reference's recovery tolerances (test/test_optimization.jl). */
/* returns model value m; interior flag; s */
double celeste_oracle_solve_tr_capped(int n, const double *g, const double *H, double delta, int secular_iters, double *s,
                                      int *interior_out);
double celeste_oracle_solve_tr(int n, const double *g, const double *H, double delta, double *s, int *interior_out) {
    return celeste_oracle_solve_tr_capped(n, g, H, delta, 20, s, interior_out);
}
/* secular_iters: cap of the Newton iterations on lambda (20 = to convergence; 5 = Optim.jl's default) */
double celeste_oracle_solve_tr_capped(int n, const double *g, const double *H, double delta, int secular_iters, double *s,
                                      int *interior_out) {
    double *w = (double *)malloc(sizeof(double) * n), *V = (double *)malloc(sizeof(double) * n * n);
    double *qg = (double *)malloc(sizeof(double) * n), *c = (double *)malloc(sizeof(double) * n);
    celeste_oracle_jacobi_eig(n, H, w, V);
    for (int i = 0; i < n; ++i) { double t = 0; for (int k = 0; k < n; ++k) t += V[k + n * i] * g[k]; qg[i] = t; }
    const double wmin = w[0], wmax = w[n - 1], d2 = delta * delta;
    int interior = 0;
    if (wmin >= 1e-8) {
        double p2 = 0; for (int i = 0; i < n; ++i) p2 += (qg[i] / w[i]) * (qg[i] / w[i]);
        if (p2 <= d2) interior = 1;
    }
    if (interior) {
        for (int i = 0; i < n; ++i) c[i] = -qg[i] / w[i];
    } else {
        const double lambda_lb = -wmin + fmax(1e-8, 1e-8 * (wmax - wmin));
        double lambda = lambda_lb;
        int hard = 0;
        if (wmin < 0) {
            int cand = 1, idx = 0;
            while (idx < n && fabs(w[0] - w[idx]) <= 1e-10) { if (fabs(qg[idx]) > 1e-10) { cand = 0; break; } ++idx; }
            if (cand) {
                double p2 = 0;
                for (int i = idx; i < n; ++i) p2 += (qg[i] / (w[i] + lambda)) * (qg[i] / (w[i] + lambda));
                if (p2 <= d2) {   /* N&W (4.45): to the boundary along the lowest eigenvector */
                    hard = 1;
                    for (int i = 0; i < n; ++i) c[i] = i < idx ? 0.0 : -qg[i] / (w[i] + lambda);
                    c[0] = sqrt(d2 - p2);
                }
            }
        }
        if (!hard) {
            for (int it = 0; it < secular_iters; ++it) {
                double q2 = 0, q3 = 0;
                for (int i = 0; i < n; ++i) { c[i] = -qg[i] / (w[i] + lambda); q2 += c[i] * c[i]; q3 += c[i] * c[i] / (w[i] + lambda); }
                const double prev = lambda;
                lambda += q2 * (sqrt(q2) - delta) / (delta * q3);
                if (lambda < lambda_lb) lambda = 0.5 * (prev - lambda_lb) + lambda_lb;
                if (fabs(lambda - prev) < 1e-10 || lambda <= prev) break;
            }
        }
    }
    double m = 0;
    for (int i = 0; i < n; ++i) m += qg[i] * c[i] + 0.5 * w[i] * c[i] * c[i];
    for (int k = 0; k < n; ++k) { double t = 0; for (int i = 0; i < n; ++i) t += V[k + n * i] * c[i]; s[k] = t; }
    if (interior_out) *interior_out = interior;
    free(w); free(V); free(qg); free(c);
    return m;
}

/* ---- maximize! (ElboMaximize.jl:228-242) for one target with frozen neighbours --------------------------- */
typedef struct celeste_optim_config_oracle {
    double loc_width, loc_scale; int32_t max_iters; int32_t include_kl;
    double xtol_abs, ftol_rel, gtol, initial_delta, delta_hat;
    int32_t tr_secular_iters, reserved;   /* 0 = to convergence (<= 20); 5 = Optim.jl's cap */
} OptCfg;

static int eval_free(const celeste_problem_t *pr, double *vp, int target, uint32_t flags, const double *x, const Boxes *b,
                     double *f, double *g, double *H) {
    double *vs = vp + (size_t)target * P;
    celeste_oracle_to_bound(x, b, vs, NULL, NULL, NULL);
    double v, d[P]; double *h = (double *)malloc(sizeof(double) * P * P);
    int st = celeste_oracle_elbo(pr, vp, target, flags, &v, d, h, NULL, NULL);
    if (st == 0) {
        propagate(x, b, d, h, g, H);
        *f = -v;
        for (int i = 0; i < NF; ++i) g[i] = -g[i];
        for (int i = 0; i < NF * NF; ++i) H[i] = -H[i];
    }
    free(h);
    return st;
}

/* vp (S x 44) is updated in place for `target`; returns status; stats: [iterations, f_evals, final elbo].
 * pos_center (may be NULL = the current position): centre of the position box, which the reference keeps where the
 * first ElboConfig of the source put it across the sweeps of joint inference (ParallelRun.jl:96-100). */
int celeste_oracle_maximize_at(const celeste_problem_t *pr, double *vp, int32_t target, const OptCfg *cfg,
                               const double *pos_center, double *stats);
int celeste_oracle_maximize(const celeste_problem_t *pr, double *vp, int32_t target, const OptCfg *cfg, double *stats) {
    return celeste_oracle_maximize_at(pr, vp, target, cfg, NULL, stats);
}
int celeste_oracle_maximize_at(const celeste_problem_t *pr, double *vp, int32_t target, const OptCfg *cfg,
                               const double *pos_center, double *stats) {
    const uint32_t flags = CELESTE_FLAG_GRAD | CELESTE_FLAG_HESS | (cfg->include_kl ? CELESTE_FLAG_KL : 0);
    double *vs = vp + (size_t)target * P;
    double centre[P]; memcpy(centre, vs, sizeof centre);
    if (pos_center) { centre[0] = pos_center[0]; centre[1] = pos_center[1]; }
    Boxes b; make_boxes(&b, centre, cfg->loc_width, cfg->loc_scale);
    enforce(vs, &b);
    double x[NF], xt[NF], s[NF], g[NF], gt[NF];
    double *H = (double *)malloc(sizeof(double) * NF * NF), *Ht = (double *)malloc(sizeof(double) * NF * NF);
    celeste_oracle_to_free(vs, &b, x);
    double f, ft, delta = cfg->initial_delta;
    int evals = 1, it = 0;
    int st = eval_free(pr, vp, target, flags, x, &b, &f, g, H);
    const int sec = cfg->tr_secular_iters > 0 ? cfg->tr_secular_iters : 20;
    if (st == 0) {   /* Optim.optimize tests the gradient at the starting point before its first iteration */
        double gmax0 = 0;
        for (int i = 0; i < NF; ++i) gmax0 = fmax(gmax0, fabs(g[i]));
        if (gmax0 <= cfg->gtol) st = -1;   /* stationary already: skip the loop (st is reset below) */
    }
    while (st == 0 && it < cfg->max_iters) {
        ++it;
        int interior;
        const double m = celeste_oracle_solve_tr_capped(NF, g, H, delta, sec, s, &interior);
        for (int i = 0; i < NF; ++i) xt[i] = x[i] + s[i];
        st = eval_free(pr, vp, target, flags, xt, &b, &ft, gt, Ht); ++evals;
        if (st != 0) break;
        double rho;
        if (fabs(m) <= 2.220446049250313e-16) rho = 1.0;
        else if (m > 0) rho = 0.25 - 1.0;
        else rho = (f - ft) / (0 - m);
        if (rho < 0.25) delta *= 0.25;
        else if (rho > 0.75 && !interior) delta = fmin(2 * delta, cfg->delta_hat);
        if (rho > 0.1) {
            double dx = 0, gmax = 0;
            for (int i = 0; i < NF; ++i) { dx = fmax(dx, fabs(xt[i] - x[i])); gmax = fmax(gmax, fabs(gt[i])); }
            const double df = fabs(ft - f);
            memcpy(x, xt, sizeof x); memcpy(g, gt, sizeof g); memcpy(H, Ht, sizeof(double) * NF * NF);
            const double fprev = f; f = ft; (void)fprev;
            if (dx <= cfg->xtol_abs || df <= cfg->ftol_rel * fabs(f) || gmax <= cfg->gtol) break;
        }
    }
    if (st == -1) st = 0;
    celeste_oracle_to_bound(x, &b, vs, NULL, NULL, NULL);
    if (stats) { stats[0] = it; stats[1] = evals; stats[2] = -f; }
    free(H); free(Ht);
    return st;
}

/* exported helpers for the tests */
void celeste_oracle_constraints_roundtrip(const double *vs_in, double loc_width, double loc_scale, double *x, double *vs_out, double *J) {
    Boxes b; make_boxes(&b, vs_in, loc_width, loc_scale);
    double vs[P]; memcpy(vs, vs_in, sizeof vs);
    enforce(vs, &b);
    celeste_oracle_to_free(vs, &b, x);
    celeste_oracle_to_bound(x, &b, vs_out, J, NULL, NULL);
}
