// elbo_kernels.h -- hand-written HIP kernels for gfx950 (CDNA4, wave64).
//
// Three kernels per batch of targets:
//   prep_kernel   one thread per (source, image, component): the reference's
//                 load_source_brightnesses + load_bvn_mixtures! (source_brightness.jl:213-229,
//                 fsm_util.jl:111-169) for every source, into SrcImg / Comp / SrcGeo tables.
//   pixel_kernel  one wavefront per (target, image, pixel chunk): the reference's
//                 elbo_likelihood pixel loop + add_pixel_term! (elbo_objective.jl:330-474).
//                 FP64-VALU bound (28 exp + ~100 flops per component per pixel); pixel / sky / iota
//                 reads are coalesced along h; per-source constants are wave-uniform, so they are
//                 fetched with scalar loads into SGPRs; per-lane sums live in VGPRs and are folded
//                 with a wave64 butterfly.  Exact reformulation: the per-pixel term only
//                 depends on the 44 parameters through 10 reduced variables, so 66 sums per patch
//                 replace the reference's 1 + 44 + 44*44.
//   lift_kernel   one workgroup per target: chain rule from the reduced space to the 44 canonical
//                 parameters (calculate_G_s! block structure, elbo_objective.jl:17-233), plus the
//                 analytic KL term (elbo_kl.jl:94-154) and the finiteness checks (elbo_args.jl:145).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include "elbo_device.h"
#include "../../include/celeste_mi355x.h"

// galaxy prototypes, normalised (light_source_model.jl:45-75), filled at context creation
__constant__ double c_eta[16];
__constant__ double c_nu[16];

// ---------------------------------------------------------------------------------------------
// prep_kernel
// ---------------------------------------------------------------------------------------------
__device__ inline void bvn_cov(double ab, double angle, double scale, double &x11, double &x12, double &x22) {
    // get_bvn_cov (BivariateNormals.jl:29-43)
    double sp, cp;
    sincos(angle, &sp, &cp);
    double ab_term = ab * ab - 1.0;
    double s2 = scale * scale;
    x12 = -s2 * cp * sp * ab_term;
    x11 = s2 * (1.0 + ab_term * (sp * sp));
    x22 = s2 * (1.0 + ab_term * (cp * cp));
}

// E_l_a[b, i], E_ll_a[b, i] values (source_brightness.jl:46-50, 123-127); b is 0-based
__device__ inline void brightness(const double *vs, int i, int b, double &El, double &Ell) {
    const double r = vs[6 + i], v = vs[8 + i];
    const double *cm = vs + 10 + 4 * i, *cv = vs + 18 + 4 * i;
    double l = r + 0.5 * v, ll = 2 * r + 2 * v;
    if (b >= 3) { l += cm[2] + .5 * cv[2]; ll += 2 * cm[2] + 2 * cv[2]; }
    if (b >= 4) { l += cm[3] + .5 * cv[3]; ll += 2 * cm[3] + 2 * cv[3]; }
    if (b <= 1) { l += -cm[1] + .5 * cv[1]; ll += -2 * cm[1] + 2 * cv[1]; }
    if (b <= 0) { l += -cm[0] + .5 * cv[0]; ll += -2 * cm[0] + 2 * cv[0]; }
    // the reference multiplies the per-colour exponentials; the product of exps is evaluated the same
    // way here so that rounding matches to ~1 ulp per factor
    double e = exp(r + 0.5 * v), ee = exp(2 * r + 2 * v);
    if (b >= 3) { e *= exp(cm[2] + .5 * cv[2]); ee *= exp(2 * cm[2] + 2 * cv[2]); }
    if (b >= 4) { e *= exp(cm[3] + .5 * cv[3]); ee *= exp(2 * cm[3] + 2 * cv[3]); }
    if (b <= 1) { e *= exp(-cm[1] + .5 * cv[1]); ee *= exp(-2 * cm[1] + 2 * cv[1]); }
    if (b <= 0) { e *= exp(-cm[0] + .5 * cv[0]); ee *= exp(-2 * cm[0] + 2 * cv[0]); }
    (void)l; (void)ll;
    El = e; Ell = ee;
}

__global__ void __launch_bounds__(64)
prep_kernel(const double *__restrict__ vp, const DevImage *__restrict__ images,
            const DevPatch *__restrict__ patches, int S, int N, int K,
            SrcImg *__restrict__ srcimg, Comp *__restrict__ comps, SrcGeo *__restrict__ geo) {
    const int NC = 14 * K;
    const int sn = blockIdx.x;  // s * N + n
    const int s = sn / N, n = sn - s * N;
    const int c = threadIdx.x;
    const double *vs = vp + (size_t)s * CEL_P;
    const DevPatch &p = patches[sn];
    const double d0 = vs[0] - p.wc[0], d1 = vs[1] - p.wc[1];
    const double m1 = p.J[0] * d0 + p.J[2] * d1 + p.pc[0];  // linear_world_to_pix (wcs_utils.jl:14-18)
    const double m2 = p.J[1] * d0 + p.J[3] * d1 + p.pc[1];
    double x11, x12, x22;
    bvn_cov(vs[3], vs[4], vs[5], x11, x12, x22);
    if (c < NC) {
        // component order of populate_gal_fsm!: type i, prototype j, psf k (k fastest)
        int i, j, k;
        if (c < 8 * K) { i = 0; j = c / K; k = c - j * K; }
        else { int cc = c - 8 * K; i = 1; j = cc / K; k = cc - j * K; }
        const double *pc = p.psf + 6 * k;
        const double nu = c_nu[8 * i + j];
        const double s11 = pc[3] + nu * x11, s12 = pc[4] + nu * x12, s22 = pc[5] + nu * x22;
        const double det = s11 * s22 - s12 * s12;
        const double idet = 1.0 / det;
        const double z = (pc[0] * c_eta[8 * i + j]) / (sqrt(det) * (2.0 * M_PI));  // BvnComponent (BivariateNormals.jl:158,185)
        Comp o;
        o.p11 = s22 * idet; o.p12 = -s12 * idet; o.p22 = s11 * idet;
        o.mu1 = pc[1] + m1; o.mu2 = pc[2] + m2;
        const double dev = vs[2];
        o.zf = z * (i == 0 ? dev : 1.0 - dev);
        o.zd = (i == 0 ? z : -z);
        o.nu = nu;
        comps[(size_t)sn * NC + c] = o;
    }
    if (c == 63) {
        const int b = images[n].band - 1;
        double El0, Ell0, El1, Ell1;
        brightness(vs, 0, b, El0, Ell0);
        brightness(vs, 1, b, El1, Ell1);
        SrcImg o;
        o.m1 = m1; o.m2 = m2;
        o.c0 = vs[26] * El0; o.c1 = vs[27] * El1;
        o.q0 = vs[26] * Ell0; o.q1 = vs[27] * Ell1;
        o.pad0 = 0; o.pad1 = 0;
        srcimg[sn] = o;
    }
    if (n == 0 && c == 62) {
        // GalaxySigmaDerivs without nuBar (BivariateNormals.jl:346-397); argument order there is
        // (angle, axis_ratio, radius); columns are (axis_ratio, angle, radius)
        SrcGeo g;
        const double ab = vs[3], ang = vs[4], r = vs[5];
        double sn_, cs_;
        sincos(ang, &sn_, &cs_);
        const double cos_sin = cs_ * sn_, sin_sq = sn_ * sn_, cos_sq = cs_ * cs_;
        const double r2 = r * r;
        double *j = g.jsh, *t = g.tsh;
        const double c1 = 2 * ab * r2;
        j[0] = c1 * sin_sq; j[1] = -c1 * cos_sin; j[2] = c1 * cos_sq;
        const double c2 = r2 * (ab * ab - 1);
        j[3] = c2 * (2 * cos_sin); j[4] = c2 * (sin_sq - cos_sq); j[5] = c2 * (-2 * cos_sin);
        j[6] = 2 * x11 / r; j[7] = 2 * x12 / r; j[8] = 2 * x22 / r;
        const double a = 2 * r2;
        t[0] = sin_sq * a; t[1] = -cos_sin * a; t[2] = cos_sq * a;
        t[3] = 2 * cos_sin * a * ab; t[4] = (sin_sq - cos_sq) * a * ab; t[5] = -2 * cos_sin * a * ab;
        t[6] = 2 * j[0] / r; t[7] = 2 * j[1] / r; t[8] = 2 * j[2] / r;
        t[9] = t[3]; t[10] = t[4]; t[11] = t[5];
        const double b2 = a * (ab * ab - 1);
        t[12] = (cos_sq - sin_sq) * b2; t[13] = 2 * cos_sin * b2; t[14] = (sin_sq - cos_sq) * b2;
        t[15] = 2 * j[3] / r; t[16] = 2 * j[4] / r; t[17] = 2 * j[5] / r;
        t[18] = t[6]; t[19] = t[7]; t[20] = t[8];
        t[21] = t[15]; t[22] = t[16]; t[23] = t[17];
        t[24] = 2 * x11 / r2; t[25] = 2 * x12 / r2; t[26] = 2 * x22 / r2;
        int fin = 1;
        for (int q = 0; q < CEL_P; ++q) fin &= (int)isfinite(vs[q]);
        g.finite = fin; g.pad = 0;
        geo[s] = g;
    }
}

// ---------------------------------------------------------------------------------------------
// pixel_kernel
// ---------------------------------------------------------------------------------------------
__device__ inline void bspline_w(double f, double w[4]) {
    const double o = 1.0 - f;
    w[0] = o * o * o * (1.0 / 6); w[1] = 2.0 / 3 - f * f + f * f * f * 0.5;
    w[2] = 2.0 / 3 - o * o + o * o * o * 0.5; w[3] = f * f * f * (1.0 / 6);
}
__device__ inline void bspline_dw(double f, double dw[4], double ddw[4]) {
    const double o = 1.0 - f;
    dw[0] = -0.5 * o * o; dw[1] = -2 * f + 1.5 * f * f; dw[2] = 2 * o - 1.5 * o * o; dw[3] = 0.5 * f * f;
    ddw[0] = o; ddw[1] = -2 + 3 * f; ddw[2] = -2 + 3 * o; ddw[3] = f;
}

// star_light_density! value only (fsm_util.jl:221-237): softpluslikeinv(itp[h - m1 + 26, w - m2 + 26])
__device__ inline double star_value(const double *__restrict__ coef, double xh, double xw) {
    int ix = (int)floor(xh); ix = ix < 1 ? 1 : (ix > 50 ? 50 : ix);
    int iy = (int)floor(xw); iy = iy < 1 ? 1 : (iy > 50 ? 50 : iy);
    double wx[4], wy[4];
    bspline_w(xh - ix, wx); bspline_w(xw - iy, wy);
    const double *c0 = coef + (ix - 1) + CEL_COEF * (iy - 1);
    double y = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const double *cb = c0 + CEL_COEF * b;
        double r = cb[0] * wx[0] + cb[1] * wx[1] + cb[2] * wx[2] + cb[3] * wx[3];
        y += r * wy[b];
    }
    return y < 0 ? 1e-3 * exp(y) : 1e-3 * (y + 1.0);
}

// value of the galaxy density sum_c zf_c exp(-0.5 d' P d) (populate_gal_fsm!, inactive branch)
__device__ inline double galaxy_value(const Comp *__restrict__ tc, int NC, double hh, double ww) {
    double v = 0;
    for (int c = 0; c < NC; ++c) {
        const Comp k = tc[c];
        const double d1 = hh - k.mu1, d2 = ww - k.mu2;
        const double py1 = k.p11 * d1 + k.p12 * d2, py2 = k.p12 * d1 + k.p22 * d2;
        v += k.zf * exp(-0.5 * (d1 * py1 + d2 * py2));
    }
    return v;
}

__device__ inline double wave_sum(double x) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

// MODE 0: value only; MODE 2: value + gradient + Hessian sums
template <int MODE>
__global__ void __launch_bounds__(64)
pixel_kernel(const DevImage *__restrict__ images, const DevPatch *__restrict__ patches,
             const double *__restrict__ coefs, const uint8_t *__restrict__ bitmaps,
             const SrcImg *__restrict__ srcimg, const Comp *__restrict__ comps,
             const int64_t *__restrict__ nbr_off, const int32_t *__restrict__ nbr_idx,
             const int32_t *__restrict__ targets, int N, int NC, int CH, int chunk_px,
             double *__restrict__ acc) {
    const int wg = blockIdx.x;
    const int ch = wg % CH;
    const int tn = wg / CH;
    const int ti = tn / N, n = tn - ti * N;
    const int t = targets[ti];
    const DevPatch &P = patches[(size_t)t * N + n];
    const int H2 = P.H2, W2 = P.W2;
    const int npx = H2 * W2;
    const int p0 = ch * chunk_px;
    if (p0 >= npx) return;  // the lift kernel recomputes this predicate
    const int p1 = min(npx, p0 + chunk_px);
    const DevImage &img = images[n];
    const int lane = threadIdx.x;
    const SrcImg si = srcimg[(size_t)t * N + n];
    const Comp *__restrict__ tc = comps + ((size_t)t * N + n) * NC;
    const double *__restrict__ tcoef = coefs + (size_t)P.stamp * (CEL_COEF * CEL_COEF);
    const int64_t nb0 = nbr_off[t], nb1 = nbr_off[t + 1];
    const double c0 = si.c0, c1 = si.c1, q0 = si.q0, q1 = si.q1;
    // index offsets of the star spline: itp[h - m1 + 26, w - m2 + 26]
    const double sh0 = 26.0 - si.m1, sw0 = 26.0 - si.m2;

    double a[ACC_N];
#pragma unroll
    for (int i = 0; i < ACC_N; ++i) a[i] = 0.0;

    for (int base = p0; base < p1; base += 64) {
        const int idx = base + lane;
        if (idx >= p1) continue;
        const int w2 = idx / H2, h2 = idx - w2 * H2;  // 0-based patch coordinates, h fastest
        const int h = P.off_h + h2 + 1, w = P.off_w + w2 + 1;  // 1-based image coordinates
        const size_t gi = (size_t)(h - 1) + (size_t)img.H * (w - 1);
        const float xf = img.pixels[gi];
        bool own_bit = true;
        if (P.bitmap_off >= 0) own_bit = bitmaps[P.bitmap_off + h2 + (int64_t)H2 * w2] != 0;
        if (!own_bit || isnan(xf)) continue;  // elbo_objective.jl:445,459
        const double hh = (double)h, ww = (double)w;
        double Ebar = (double)img.sky[gi];  // epsilon + neighbours
        double Vbar = 0.0;
        double n_inact = 0.0;

        // ---- neighbours: value-only contributions (is_active_source == false) ----
        for (int64_t q = nb0; q < nb1; ++q) {
            const int s2 = nbr_idx[q];
            const DevPatch &Q = patches[(size_t)s2 * N + n];
            const int ph2 = h - Q.off_h, pw2 = w - Q.off_w;  // 1-based in the neighbour's patch
            bool in = (ph2 >= 1) & (ph2 <= Q.H2) & (pw2 >= 1) & (pw2 < Q.W2);  // strict: elbo_objective.jl:349
            if (in && Q.bitmap_off >= 0) in = bitmaps[Q.bitmap_off + (ph2 - 1) + (int64_t)Q.H2 * (pw2 - 1)] != 0;
            const SrcImg sj = srcimg[(size_t)s2 * N + n];  // wave-uniform
            if (in) {
                const double f0 = star_value(coefs + (size_t)Q.stamp * (CEL_COEF * CEL_COEF),
                                             hh + (26.0 - sj.m1), ww + (26.0 - sj.m2));
                const double f1 = galaxy_value(comps + ((size_t)s2 * N + n) * NC, NC, hh, ww);
                const double En = sj.c0 * f0 + sj.c1 * f1;            // E_G_s.v  (elbo_objective.jl:62-65)
                const double E2n = sj.q0 * (f0 * f0) + sj.q1 * (f1 * f1);
                Ebar += En;
                Vbar += E2n - En * En;                                  // var_G_s.v (elbo_objective.jl:204)
                n_inact += 1.0;
            }
        }

        // ---- the active source ----
        const bool own = (w2 < W2 - 1);  // 1 <= w2 < W2 (1-based), elbo_objective.jl:349
        double f0 = 0, f1 = 0;
        double f0g[2] = {0, 0}, f0h[3] = {0, 0, 0};   // d/dm, d2/dm2 (m1m1, m1m2, m2m2)
        double g1[6] = {0, 0, 0, 0, 0, 0};            // m1 m2 dev Xi11 Xi12 Xi22
        double h1[21];
#pragma unroll
        for (int i = 0; i < 21; ++i) h1[i] = 0.0;
        if (own) {
            // star: spline value + derivatives with respect to the index, then index = h - m + 26
            {
                const double xh = hh + sh0, xw = ww + sw0;
                int ix = (int)floor(xh); ix = ix < 1 ? 1 : (ix > 50 ? 50 : ix);
                int iy = (int)floor(xw); iy = iy < 1 ? 1 : (iy > 50 ? 50 : iy);
                const double fx = xh - ix, fy = xw - iy;
                double wx[4], wy[4];
                bspline_w(fx, wx); bspline_w(fy, wy);
                const double *cc = tcoef + (ix - 1) + CEL_COEF * (iy - 1);
                if (MODE == 0) {
                    double y = 0;
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const double *cb = cc + CEL_COEF * b;
                        y += (cb[0] * wx[0] + cb[1] * wx[1] + cb[2] * wx[2] + cb[3] * wx[3]) * wy[b];
                    }
                    f0 = y < 0 ? 1e-3 * exp(y) : 1e-3 * (y + 1.0);
                } else {
                    double dwx[4], ddwx[4], dwy[4], ddwy[4];
                    bspline_dw(fx, dwx, ddwx); bspline_dw(fy, dwy, ddwy);
                    double y = 0, yx = 0, yy = 0, yxx = 0, yxy = 0, yyy = 0;
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const double *cb = cc + CEL_COEF * b;
                        const double k0 = cb[0], k1 = cb[1], k2 = cb[2], k3 = cb[3];
                        const double r = k0 * wx[0] + k1 * wx[1] + k2 * wx[2] + k3 * wx[3];
                        const double rx = k0 * dwx[0] + k1 * dwx[1] + k2 * dwx[2] + k3 * dwx[3];
                        const double rxx = k0 * ddwx[0] + k1 * ddwx[1] + k2 * ddwx[2] + k3 * ddwx[3];
                        y += r * wy[b]; yx += rx * wy[b]; yxx += rxx * wy[b];
                        yy += r * dwy[b]; yxy += rx * dwy[b]; yyy += r * ddwy[b];
                    }
                    // softpluslikeinv and its derivatives; not C2 at 0, branch exactly (fsm_util.jl:222)
                    double gv, gp, gpp;
                    if (y < 0) { gv = 1e-3 * exp(y); gp = gv; gpp = gv; }
                    else { gv = 1e-3 * (y + 1.0); gp = 1e-3; gpp = 0.0; }
                    f0 = gv;
                    // d(index)/dm = -1
                    const double ym1 = -yx, ym2 = -yy;
                    f0g[0] = gp * ym1; f0g[1] = gp * ym2;
                    f0h[0] = gpp * ym1 * ym1 + gp * yxx;
                    f0h[1] = gpp * ym1 * ym2 + gp * yxy;
                    f0h[2] = gpp * ym2 * ym2 + gp * yyy;
                }
            }
            // galaxy: 14 * psf_K bivariate normals (accum_galaxy_pos!, fsm_util.jl:255-346), with the
            // per-component (x, Sigma) -> (pos, shape) transforms hoisted out of the pixel loop
            for (int c = 0; c < NC; ++c) {
                const Comp k = tc[c];
                const double d1 = hh - k.mu1, d2 = ww - k.mu2;
                const double py1 = k.p11 * d1 + k.p12 * d2, py2 = k.p12 * d1 + k.p22 * d2;
                const double e = exp(-0.5 * (d1 * py1 + d2 * py2));   // eval_bvn_pdf!
                const double f = k.zf * e;
                f1 += f;
                if (MODE != 0) {
                    const double fd = k.zd * e;
                    const double fn = f * k.nu;
                    const double aa = py1 * py1, ab = py1 * py2, bb = py2 * py2;
                    // get_bvn_derivs!: bvn_sig_d (BivariateNormals.jl:266-271)
                    const double sd1 = 0.5 * (aa - k.p11), sd2 = ab - k.p12, sd3 = 0.5 * (bb - k.p22);
                    g1[0] += f * py1; g1[1] += f * py2;   // d/dm = -d/dx = +py
                    g1[2] += fd;
                    g1[3] += fn * sd1; g1[4] += fn * sd2; g1[5] += fn * sd3;
                    // (m, m): f (py py' - P)
                    h1[0] += f * (aa - k.p11); h1[1] += f * (ab - k.p12); h1[6] += f * (bb - k.p22);
                    // (m, dev)
                    h1[2] += fd * py1; h1[7] += fd * py2;
                    // (m, Xi): -(xsig_h[x, sg] - py_x sd[sg]) * nu
                    h1[3] += fn * (py1 * sd1 - py1 * k.p11);
                    h1[4] += fn * (py1 * sd2 - (py1 * k.p12 + py2 * k.p11));
                    h1[5] += fn * (py1 * sd3 - py2 * k.p12);
                    h1[8] += fn * (py2 * sd1 - py1 * k.p12);
                    h1[9] += fn * (py2 * sd2 - (py1 * k.p22 + py2 * k.p12));
                    h1[10] += fn * (py2 * sd3 - py2 * k.p22);
                    // (dev, Xi)
                    const double fdn = fd * k.nu;
                    h1[12] += fdn * sd1; h1[13] += fdn * sd2; h1[14] += fdn * sd3;
                    // (Xi, Xi): nu^2 f (sigsig_h + sd sd')
                    const double fnn = fn * k.nu;
                    h1[15] += fnn * (sd1 * sd1 - aa * k.p11 + 0.5 * k.p11 * k.p11);
                    h1[16] += fnn * (sd1 * sd2 - ab * k.p11 - aa * k.p12 + k.p11 * k.p12);
                    h1[17] += fnn * (sd1 * sd3 - ab * k.p12 + 0.5 * k.p12 * k.p12);
                    h1[18] += fnn * (sd2 * sd2 - aa * k.p22 - 2.0 * ab * k.p12 - bb * k.p11 + k.p11 * k.p22 + k.p12 * k.p12);
                    h1[19] += fnn * (sd2 * sd3 - ab * k.p22 - bb * k.p12 + k.p22 * k.p12);
                    h1[20] += fnn * (sd3 * sd3 - bb * k.p22 + 0.5 * k.p22 * k.p22);
                }
            }
        }
        // h1 packed upper triangle of the 6x6 (m1 m2 dev Xi11 Xi12 Xi22):
        // row0: 0..5, row1: 6..10, row2: 11..14 (11 = dev,dev = 0), row3: 15..17, row4: 18..19, row5: 20

        // ---- per-pixel term (add_pixel_term!, add_elbo_log_term!) ----
        const double A = c0 * f0 + c1 * f1;                       // E_G_s.v
        const double B = q0 * (f0 * f0) + q1 * (f1 * f1);         // E_G2_s.v
        const double E = Ebar + A;                                // E_G.v
        const double V = Vbar + (B - A * A);                      // var_G.v
        const double x = (double)xf;
        const float iota_f = img.iota[h - 1];
        const double iota = (double)iota_f;
        // log(iota) is a Float32 log in the reference (iota::Float32, elbo_objective.jl:292)
        const double log_iota = (double)(float)log(iota);
        const double iE = 1.0 / E;
        const double iE2 = iE * iE;
        a[0] += x * (log_iota + (log(E) - V * (0.5 * iE2))) - iota * E - lgamma(x + 1.0);
        a[ACC_CNT] += own ? 1.0 : 0.0;
        a[ACC_CNT + 1] += n_inact;
        if (MODE != 0 && own) {
            const double iE3 = iE2 * iE;
            const double w1 = x * (iE + V * iE3) - iota;          // dT/dE
            const double w2 = -0.5 * x * iE2;                     // dT/dVar
            const double w11 = -x * (iE2 + 3.0 * V * iE2 * iE2);  // d2T/dE2
            const double w12 = x * iE3;                           // d2T/dE dVar
            const double alpha = w1 - 2.0 * A * w2;
            const double beta = w11 - 2.0 * w2 - 4.0 * A * w12;
            // geometry derivatives of A and B: index g = 0..5 <-> reduced variable 4 + g
            double dAg[6], dBg[6];
            dAg[0] = c0 * f0g[0] + c1 * g1[0]; dAg[1] = c0 * f0g[1] + c1 * g1[1];
            dBg[0] = 2.0 * (q0 * f0 * f0g[0] + q1 * f1 * g1[0]);
            dBg[1] = 2.0 * (q0 * f0 * f0g[1] + q1 * f1 * g1[1]);
#pragma unroll
            for (int g = 2; g < 6; ++g) { dAg[g] = c1 * g1[g]; dBg[g] = 2.0 * q1 * f1 * g1[g]; }
            // gradient
            a[1] += alpha * f0; a[2] += alpha * f1;
            a[3] += w2 * f0 * f0; a[4] += w2 * f1 * f1;
#pragma unroll
            for (int g = 0; g < 6; ++g) a[5 + g] += alpha * dAg[g] + w2 * dBg[g];
            // Hessian: (c, c)
            a[hidx(0, 0)] += beta * f0 * f0; a[hidx(0, 1)] += beta * f0 * f1; a[hidx(1, 1)] += beta * f1 * f1;
            // (c, q)
            a[hidx(0, 2)] += w12 * f0 * (f0 * f0); a[hidx(0, 3)] += w12 * f0 * (f1 * f1);
            a[hidx(1, 2)] += w12 * f1 * (f0 * f0); a[hidx(1, 3)] += w12 * f1 * (f1 * f1);
            // (c, geo) and (q, geo)
            const double f0gx[6] = {f0g[0], f0g[1], 0, 0, 0, 0};
#pragma unroll
            for (int g = 0; g < 6; ++g) {
                const double u = beta * dAg[g] + w12 * dBg[g];
                a[hidx(0, 4 + g)] += alpha * f0gx[g] + f0 * u;
                a[hidx(1, 4 + g)] += alpha * g1[g] + f1 * u;
                a[hidx(2, 4 + g)] += 2.0 * w2 * f0 * f0gx[g] + w12 * (f0 * f0) * dAg[g];
                a[hidx(3, 4 + g)] += 2.0 * w2 * f1 * g1[g] + w12 * (f1 * f1) * dAg[g];
            }
            // (geo, geo)
            const double k1 = alpha * c1 + 2.0 * w2 * q1 * f1;    // multiplies d2 f1
            const double k0 = alpha * c0 + 2.0 * w2 * q0 * f0;    // multiplies d2 f0
            const double r1 = 2.0 * w2 * q1, r0 = 2.0 * w2 * q0;  // multiply df df'
            int hp = 0;
#pragma unroll
            for (int g = 0; g < 6; ++g) {
#pragma unroll
                for (int g2 = g; g2 < 6; ++g2, ++hp) {
                    double v = k1 * h1[hp] + r1 * g1[g] * g1[g2] + beta * dAg[g] * dAg[g2] +
                               w12 * (dAg[g] * dBg[g2] + dBg[g] * dAg[g2]);
                    if (g2 < 2) {
                        const int sp = g + g2;  // (0,0)->0 (0,1)->1 (1,1)->2
                        v += k0 * f0h[sp] + r0 * f0g[g] * f0g[g2];
                    }
                    a[hidx(4 + g, 4 + g2)] += v;
                }
            }
        }
    }

    // ---- wave64 butterfly, one 68-double record per (target, image, chunk) ----
    double *__restrict__ out = acc + (size_t)wg * ACC_N;
#pragma unroll
    for (int i = 0; i < ACC_N; ++i) {
        if (MODE == 0 && i > 0 && i < ACC_CNT) continue;
        const double s = wave_sum(a[i]);
        if (lane == (i & 63)) out[i] = s;
    }
}

// ---------------------------------------------------------------------------------------------
// lift_kernel: reduced 10-variable space -> 44 canonical parameters, + KL, + checks
// ---------------------------------------------------------------------------------------------
struct PriorDev {
    celeste_prior_t p;
    double inv_cov[2][8][16];  // inv(prior.color_cov[:, :, d, i]), column-major
    double logdet[2][8];
};

// rows of the reduced Jacobian touched by canonical parameter p: start, count, stride; class id
__device__ inline void param_rows(int p, int &start, int &cnt, int &stride, int &cls) {
    if (p < 2) { start = 4; cnt = 2; stride = 1; cls = 0; }
    else if (p == 2) { start = 6; cnt = 1; stride = 1; cls = 1; }
    else if (p < 6) { start = 7; cnt = 3; stride = 1; cls = 2; }
    else if (p < 28) {
        int i;
        if (p < 10) i = (p - 6) & 1;          // flux_loc 6,7  flux_scale 8,9
        else if (p < 26) i = ((p - 10) >> 2) & 1;  // color_mean 10..17, color_var 18..25
        else i = p - 26;                       // is_star
        start = i; cnt = 2; stride = 2; cls = 3 + i;
    } else { start = 0; cnt = 0; stride = 1; cls = 5; }
}

// index of canonical parameter p inside its type's brightness vector (bids order), -1 for is_star
__device__ inline int bright_slot(int p) {
    if (p < 8) return 0;
    if (p < 10) return 1;
    if (p < 18) return 2 + ((p - 10) & 3);
    if (p < 26) return 6 + ((p - 18) & 3);
    return -1;
}

__global__ void __launch_bounds__(256)
lift_kernel(const double *__restrict__ vp, const DevImage *__restrict__ images,
            const DevPatch *__restrict__ patches, const SrcGeo *__restrict__ geo,
            const int64_t *__restrict__ nbr_off, const int32_t *__restrict__ nbr_idx,
            const int32_t *__restrict__ targets, const double *__restrict__ acc,
            const PriorDev *__restrict__ prior, int N, int CH, int chunk_px, uint32_t flags,
            double *__restrict__ out_v, double *__restrict__ out_d, double *__restrict__ out_h,
            int64_t *__restrict__ out_cnt, int32_t *__restrict__ out_status) {
    __shared__ double sh_h[CEL_P * CEL_P];
    __shared__ double sh_d[CEL_P];
    __shared__ double sh_rec[ACC_N];
    __shared__ double sh_jz[ZV * CEL_P];
    __shared__ double sh_kap[10], sh_lam[10];
    __shared__ double sh_El[2], sh_Ell[2];
    __shared__ double sh_v, sh_cnt[2];
    __shared__ double sh_m[16], sh_t[16], sh_Ld[16][4], sh_ml[16][4];
    __shared__ int sh_bad;

    const int ti = blockIdx.x, tid = threadIdx.x;
    const int t = targets[ti];
    const double *vs = vp + (size_t)t * CEL_P;
    const bool want_grad = (flags & (CELESTE_FLAG_GRAD | CELESTE_FLAG_HESS)) != 0;
    const bool want_hess = (flags & CELESTE_FLAG_HESS) != 0;
    const SrcGeo &G = geo[t];

    for (int k = tid; k < CEL_P * CEL_P; k += blockDim.x) sh_h[k] = 0.0;
    if (tid < CEL_P) sh_d[tid] = 0.0;
    if (tid == 0) { sh_v = 0.0; sh_cnt[0] = 0.0; sh_cnt[1] = 0.0; sh_bad = 0; }
    __syncthreads();

    for (int n = 0; n < N; ++n) {
        const DevPatch &P = patches[(size_t)t * N + n];
        const int npx = P.H2 * P.W2;
        const int b = images[n].band - 1;
        // sum the chunk records of this (target, image)
        if (tid < ACC_N) {
            double s = 0.0;
            for (int ch = 0; ch < CH; ++ch)
                if (ch * chunk_px < npx) s += acc[((size_t)(ti * N + n) * CH + ch) * ACC_N + tid];
            sh_rec[tid] = s;
        }
        if (tid >= 64 + 8 && tid < 64 + 8 + 10) {
            // exponent coefficients of E_l_a[b, .] (kappa) and E_ll_a[b, .] (lambda), bids order
            const int q = tid - 72;
            double kap = 0, lam = 0;
            if (q == 0) { kap = 1; lam = 2; }
            else if (q == 1) { kap = .5; lam = 2; }
            else {
                const int c = (q - 2) & 3;
                const bool is_var = q >= 6;
                bool on; double sgn;
                if (c == 2) { on = b >= 3; sgn = 1; }
                else if (c == 3) { on = b >= 4; sgn = 1; }
                else if (c == 1) { on = b <= 1; sgn = -1; }
                else { on = b <= 0; sgn = -1; }
                if (on) { kap = is_var ? .5 : sgn; lam = is_var ? 2 : 2 * sgn; }
            }
            sh_kap[q] = kap; sh_lam[q] = lam;
        }
        if (tid >= 96 && tid < 98) {
            double El, Ell;
            brightness(vs, tid - 96, b, El, Ell);
            sh_El[tid - 96] = El; sh_Ell[tid - 96] = Ell;
        }
        __syncthreads();
        if (tid == 0) { sh_v += sh_rec[0]; sh_cnt[0] += sh_rec[ACC_CNT]; sh_cnt[1] += sh_rec[ACC_CNT + 1]; }
        if (want_grad) {
            // dense 10 x 44 Jacobian of the reduced variables for this image
            for (int k = tid; k < ZV * CEL_P; k += blockDim.x) {
                const int r = k / CEL_P, p = k - r * CEL_P;
                double v = 0.0;
                if (r >= 4 && r < 6) { if (p < 2) v = P.J[(r - 4) + 2 * p]; }
                else if (r == 6) { if (p == 2) v = 1.0; }
                else if (r >= 7) { if (p >= 3 && p < 6) v = G.jsh[(r - 7) + 3 * (p - 3)]; }
                else {
                    const int i = r & 1;          // rows 0,1 = c_i; rows 2,3 = q_i
                    const bool isq = r >= 2;
                    int st, cn, sd, cls;
                    param_rows(p, st, cn, sd, cls);
                    if (cls == 3 + i) {
                        const int slot = bright_slot(p);
                        const double Ev = isq ? sh_Ell[i] : sh_El[i];
                        if (slot < 0) v = Ev;
                        else v = vs[26 + i] * Ev * (isq ? sh_lam[slot] : sh_kap[slot]);
                    }
                }
                sh_jz[k] = v;
            }
            __syncthreads();
            if (tid < CEL_P) {
                int st, cn, sd, cls;
                param_rows(tid, st, cn, sd, cls);
                double s = 0.0;
                for (int a = 0; a < cn; ++a) { const int r = st + a * sd; s += sh_jz[r * CEL_P + tid] * sh_rec[1 + r]; }
                sh_d[tid] += s;
            }
            if (want_hess) {
                for (int k = tid; k < CEL_P * CEL_P; k += blockDim.x) {
                    const int p2 = k / CEL_P, p1 = k - p2 * CEL_P;
                    if (p1 > p2) continue;
                    int st1, cn1, sd1, cls1, st2, cn2, sd2, cls2;
                    param_rows(p1, st1, cn1, sd1, cls1);
                    param_rows(p2, st2, cn2, sd2, cls2);
                    if (cn1 == 0 || cn2 == 0) continue;
                    double s = 0.0;
                    for (int a = 0; a < cn1; ++a) {
                        const int r1 = st1 + a * sd1;
                        const double j1 = sh_jz[r1 * CEL_P + p1];
                        double inner = 0.0;
                        for (int c = 0; c < cn2; ++c) {
                            const int r2 = st2 + c * sd2;
                            const int lo = r1 < r2 ? r1 : r2, hi = r1 < r2 ? r2 : r1;
                            inner += sh_rec[hidx(lo, hi)] * sh_jz[r2 * CEL_P + p2];
                        }
                        s += j1 * inner;
                    }
                    // second derivatives of the reduced variables
                    if (cls1 == 2 && cls2 == 2) {
                        for (int sg = 0; sg < 3; ++sg) s += sh_rec[1 + 7 + sg] * G.tsh[sg + 3 * (p1 - 3) + 9 * (p2 - 3)];
                    } else if (cls1 == cls2 && cls1 >= 3 && cls1 <= 4) {
                        const int i = cls1 - 3;
                        const int s1 = bright_slot(p1), s2 = bright_slot(p2);
                        const double gc = sh_rec[1 + i], gq = sh_rec[1 + 2 + i];
                        const double El = sh_El[i], Ell = sh_Ell[i], ai = vs[26 + i];
                        if (s1 >= 0 && s2 >= 0)
                            s += ai * (gc * El * sh_kap[s1] * sh_kap[s2] + gq * Ell * sh_lam[s1] * sh_lam[s2]);
                        else if (s1 >= 0) s += gc * El * sh_kap[s1] + gq * Ell * sh_lam[s1];
                        else if (s2 >= 0) s += gc * El * sh_kap[s2] + gq * Ell * sh_lam[s2];
                    }
                    sh_h[p1 + CEL_P * p2] += s;
                }
            }
        }
        __syncthreads();
    }

    // ---- KL (subtract_kl, elbo_kl.jl:94-154), analytic derivatives ----
    if (flags & CELESTE_FLAG_KL) {
        const celeste_prior_t &pr = prior->p;
        if (tid < 16) {
            const int i = tid >> 3, d = tid & 7;
            const double k = vs[28 + 8 * i + d];
            sh_t[tid] = log(k) - log(pr.k[i][d]);
            double diff[4], tr = 0, sl = 0, quad = 0;
            for (int c = 0; c < 4; ++c) diff[c] = pr.color_mean[i][d][c] - vs[10 + 4 * i + c];
            for (int c = 0; c < 4; ++c) {
                const double lam = vs[18 + 4 * i + c];
                tr += prior->inv_cov[i][d][c + 4 * c] * lam; sl += log(lam);
            }
            for (int r = 0; r < 4; ++r) {
                double Ld = 0;
                for (int c = 0; c < 4; ++c) Ld += prior->inv_cov[i][d][r + 4 * c] * diff[c];
                sh_Ld[tid][r] = Ld; quad += diff[r] * Ld;
                sh_ml[tid][r] = 0.5 * (prior->inv_cov[i][d][r + 4 * r] - 1.0 / vs[18 + 4 * i + r]);
            }
            sh_m[tid] = 0.5 * ((tr - 4.0) + quad + (prior->logdet[i][d] - sl));
        }
        __syncthreads();
        if (tid == 0) {
            double kl = 0;
            for (int i = 0; i < 2; ++i) {
                const int ia = 26 + i;
                const double a = vs[ia];
                const double ta = log(a) - log(pr.is_star[i]);
                kl -= a * ta;
                sh_d[ia] -= ta + 1.0;
                sh_h[ia + CEL_P * ia] -= 1.0 / a;
                double ck = 0, cm = 0;
                for (int d = 0; d < 8; ++d) {
                    const int ik = 28 + 8 * i + d, q = 8 * i + d;
                    const double k = vs[ik];
                    ck += k * sh_t[q]; cm += k * sh_m[q];
                    sh_d[ik] -= a * (sh_t[q] + 1.0) + a * sh_m[q];
                    sh_h[ik + CEL_P * ik] -= a / k;
                    sh_h[ia + CEL_P * ik] -= (sh_t[q] + 1.0) + sh_m[q];
                }
                const int ir = 6 + i, iv = 8 + i;
                const double mu1 = vs[ir], var1 = vs[iv], mu2 = pr.flux_mean[i], var2 = pr.flux_var[i];
                const double g = .5 * (log(var2) - log(var1) + (var1 + (mu1 - mu2) * (mu1 - mu2)) / var2 - 1.0);
                const double g_r = (mu1 - mu2) / var2, g_v = .5 * (-1.0 / var1 + 1.0 / var2);
                kl -= a * (ck + g + cm);
                sh_d[ia] -= ck + g + cm;
                sh_d[ir] -= a * g_r; sh_d[iv] -= a * g_v;
                sh_h[ir + CEL_P * ia] -= g_r; sh_h[iv + CEL_P * ia] -= g_v;
                sh_h[ir + CEL_P * ir] -= a / var2; sh_h[iv + CEL_P * iv] -= a * .5 / (var1 * var1);
                for (int c = 0; c < 4; ++c) {
                    const int im = 10 + 4 * i + c, il = 18 + 4 * i + c;
                    const double lam = vs[il];
                    double s_mu = 0, s_l = 0, s_k = 0;
                    for (int d = 0; d < 8; ++d) {
                        const int ik = 28 + 8 * i + d, q = 8 * i + d;
                        const double k = vs[ik];
                        const double m_mu = -sh_Ld[q][c], m_l = sh_ml[q][c];
                        s_mu += k * m_mu; s_l += k * m_l; s_k += k;
                        sh_h[im + CEL_P * ik] -= a * m_mu;   // im < ik: upper triangle
                        sh_h[il + CEL_P * ik] -= a * m_l;
                    }
                    sh_d[im] -= a * s_mu; sh_d[il] -= a * s_l;
                    sh_h[im + CEL_P * ia] -= s_mu; sh_h[il + CEL_P * ia] -= s_l;
                    sh_h[il + CEL_P * il] -= a * s_k * 0.5 / (lam * lam);
                    for (int c2 = 0; c2 <= c; ++c2) {
                        const int im2 = 10 + 4 * i + c2;
                        double s = 0;
                        for (int d = 0; d < 8; ++d) s += vs[28 + 8 * i + d] * prior->inv_cov[i][d][c2 + 4 * c];
                        sh_h[im2 + CEL_P * im] -= a * s;
                    }
                }
            }
            const double x = vs[5], mu = pr.gal_radius_px_mean, s2 = pr.gal_radius_px_var;
            kl += -0.5 * (log(2.0 * M_PI) + log(s2) + (x - mu) * (x - mu) / s2);
            sh_d[5] += -(x - mu) / s2;
            sh_h[5 + CEL_P * 5] += -1.0 / s2;
            sh_v += kl;
        }
        __syncthreads();
    }

    // ---- finiteness (elbo_objective.jl:487,490), symmetrise, store ----
    int bad = 0;
    if (tid == 0 && !isfinite(sh_v)) bad = 1;
    if (want_grad && tid < CEL_P && !isfinite(sh_d[tid])) bad = 1;
    if (want_hess)
        for (int k = tid; k < CEL_P * CEL_P; k += blockDim.x) {
            const int p2 = k / CEL_P, p1 = k - p2 * CEL_P;
            if (p1 <= p2 && !isfinite(sh_h[k])) bad = 1;
        }
    if (bad) atomicOr(&sh_bad, 1);
    __syncthreads();
    if (tid == 0) {
        int st = sh_bad ? CELESTE_ERR_NONFINITE_RESULT : CELESTE_OK;
        int fin = G.finite;
        for (int64_t q = nbr_off[t]; q < nbr_off[t + 1]; ++q) fin &= geo[nbr_idx[q]].finite;
        if (!fin) st = CELESTE_ERR_NONFINITE_INPUT;
        out_status[ti] = st;
        out_v[ti] = sh_v;
        if (out_cnt) { out_cnt[2 * ti] = (int64_t)(sh_cnt[0] + 0.5); out_cnt[2 * ti + 1] = (int64_t)(sh_cnt[1] + 0.5); }
    }
    if (want_grad && out_d && tid < CEL_P) out_d[(size_t)ti * CEL_P + tid] = sh_d[tid];
    if (want_hess && out_h)
        for (int k = tid; k < CEL_P * CEL_P; k += blockDim.x) {
            const int p2 = k / CEL_P, p1 = k - p2 * CEL_P;
            out_h[(size_t)ti * CEL_P * CEL_P + k] = (p1 <= p2) ? sh_h[k] : sh_h[p2 + CEL_P * p1];
        }
}

// ---------------------------------------------------------------------------------------------
// psf_raster_kernel: get_psf_at_point (PSF.jl:150-161)
// ---------------------------------------------------------------------------------------------
__global__ void psf_raster_kernel(const double *__restrict__ psf, int K, const double *__restrict__ rows, int nr,
                                  const double *__restrict__ cols, int nc, double *__restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nr * nc) return;
    const int c = idx / nr, r = idx - c * nr;
    double s = 0;
    for (int k = 0; k < K; ++k) {
        const double *pc = psf + 6 * k;
        const double x0 = rows[r] - pc[1], x1 = cols[c] - pc[2];
        const double det = pc[3] * pc[5] - pc[4] * pc[4];
        const double i11 = pc[5] / det, i12 = -pc[4] / det, i22 = pc[3] / det;
        const double q = x0 * (i11 * x0 + i12 * x1) + x1 * (i12 * x0 + i22 * x1);
        s += pc[0] * exp(-0.5 * q - 0.5 * log(det)) / (2.0 * M_PI);
    }
    out[idx] = s;
}
