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

// Mutation testing (tests/test_mutants.py builds the library with -DCELESTE_MUTANT=k and checks that the parity
// tests notice): 0 = the product; 1 = iota indexed by column instead of row; 2 = sky plane read transposed;
// 3 = every patch uses stamp 0.  Never defined in the shipped build.
#ifndef CELESTE_MUTANT
#define CELESTE_MUTANT 0
#endif

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

// the tables of one (source, image) visit: lane c < NC computes component c, lane 63 the brightness moments.
// vs: the source's 44 parameters; p: its patch in the image; b: the image's band, 0-based; the results go to *si_out and
// comps_out[c] -- the per-visit tables in HBM (prep_kernel) or a workgroup's LDS copy (optim_fused_kernel)
// linear_world_to_pix (wcs_utils.jl:14-18)
__device__ __forceinline__ void world_to_pix(const double *__restrict__ vs, const DevPatch &p, double &m1, double &m2) {
    const double d0 = vs[0] - p.wc[0], d1 = vs[1] - p.wc[1];
    m1 = p.J[0] * d0 + p.J[2] * d1 + p.pc[0];
    m2 = p.J[1] * d0 + p.J[3] * d1 + p.pc[1];
}
// MOMENTS = false: the components only (the fused optimiser kernel forms the brightness moments on other lanes,
// brightness_moments_wave)
template <bool MOMENTS = true>
__device__ __forceinline__ void prep_visit_values(int c, const double *__restrict__ vs, const DevPatch &p, int b, int K,
                                                  SrcImg *__restrict__ si_out, Comp *__restrict__ comps_out) {
    const int NC = 14 * K;
    double m1, m2;
    world_to_pix(vs, p, m1, m2);
    double x11, x12, x22;
    bvn_cov(vs[3], vs[4], vs[5], x11, x12, x22);
    if (c < NC) {
        // component order: type i, psf component k, prototype j (j fastest) -- the 8 / 6 prototypes that share a PSF
        // component (hence its offset xiBar_k) are consecutive, so the pixel kernel forms pixel - mean once per run
        // (populate_gal_fsm! loops i, j, k; the sum does not care)
        int i, j, k;
        if (c < 8 * K) { i = 0; k = c / 8; j = c - 8 * k; }
        else { int cc = c - 8 * K; i = 1; k = cc / 6; j = cc - 6 * k; }
        const double *pc = p.psf + 6 * k;
        const double nu = c_nu[8 * i + j];
        const double s11 = pc[3] + nu * x11, s12 = pc[4] + nu * x12, s22 = pc[5] + nu * x22;
        const double det = s11 * s22 - s12 * s12;
        const double idet = 1.0 / det;
        const double z = (pc[0] * c_eta[8 * i + j]) / (sqrt(det) * (2.0 * M_PI));  // BvnComponent (BivariateNormals.jl:158,185)
        Comp o;
        o.p11 = s22 * idet; o.p12 = -s12 * idet; o.p22 = s11 * idet;
        o.xi1 = pc[1]; o.xi2 = pc[2];
        const double dev = vs[2];
        o.w0 = z * (i == 0 ? dev : 1.0 - dev);
        o.wd = (i == 0 ? z : -z);
        o.nu = nu;
        comps_out[c] = o;
    }
    if (MOMENTS && c == 63) {
        double El0, Ell0, El1, Ell1;
        brightness(vs, 0, b, El0, Ell0);
        brightness(vs, 1, b, El1, Ell1);
        SrcImg o;
        o.m1 = m1; o.m2 = m2;
        o.c0 = vs[26] * El0; o.c1 = vs[27] * El1;
        o.q0 = vs[26] * Ell0; o.q1 = vs[27] * Ell1;
        o.dev = vs[2]; o.pad1 = 0;
        *si_out = o;
    }
}

// The SrcImg of a visit by one whole wavefront: brightness() evaluates up to ten exponentials per type one after the
// other on a single lane (1.3 us on the critical path of every chunk item of the fused optimiser kernel); here lane
// 10 i + 5 w + f evaluates factor f (0: flux, 1..4: colours 3, 4, 2, 1 -- brightness()'s order) of E_l_a (w = 0) or
// E_ll_a (w = 1) of type i, and lanes 0..3 form the four products in brightness()'s order.  Every factor's argument is a
// sum of two terms scaled by powers of two, i.e. rounded once however it is contracted: the results are brightness()'s, bit
// for bit.
__device__ __forceinline__ void brightness_moments_wave(int lane, const double *__restrict__ vs, const DevPatch &p, int b,
                                                        SrcImg *__restrict__ si_out) {
    double fac = 1.0;
    if (lane < 20) {
        const int i = lane / 10, w = (lane / 5) & 1, f = lane % 5;
        const double r = vs[6 + i], v = vs[8 + i];
        const double *cm = vs + 10 + 4 * i, *cv = vs + 18 + 4 * i;
        const int col = f == 1 ? 2 : (f == 2 ? 3 : (f == 3 ? 1 : 0));            // colour index of factors 1..4
        const double sgn = f >= 3 ? -1.0 : 1.0;                                     // colours 1, 2 enter with a minus sign
        double arg;
        if (f == 0) arg = w ? 2 * r + 2 * v : r + 0.5 * v;
        else arg = w ? 2 * (sgn * cm[col]) + 2 * cv[col] : sgn * cm[col] + .5 * cv[col];
        fac = exp(arg);
    }
    const int base = lane < 4 ? 5 * lane : 0;                                       // lane q < 4: type q / 2, moment q % 2
    double prod = __shfl(fac, base, 64);
    const double f1 = __shfl(fac, base + 1, 64), f2 = __shfl(fac, base + 2, 64), f3 = __shfl(fac, base + 3, 64),
                 f4 = __shfl(fac, base + 4, 64);
    if (b >= 3) prod *= f1;
    if (b >= 4) prod *= f2;
    if (b <= 1) prod *= f3;
    if (b <= 0) prod *= f4;
    const double El0 = __shfl(prod, 0, 64), Ell0 = __shfl(prod, 1, 64), El1 = __shfl(prod, 2, 64), Ell1 = __shfl(prod, 3, 64);
    if (lane == 0) {
        SrcImg o;
        world_to_pix(vs, p, o.m1, o.m2);
        o.c0 = vs[26] * El0; o.c1 = vs[27] * El1;
        o.q0 = vs[26] * Ell0; o.q1 = vs[27] * Ell1;
        o.dev = vs[2]; o.pad1 = 0;
        *si_out = o;
    }
}
__device__ __forceinline__ void prep_visit(int c, int s, int n, int sn, const double *__restrict__ vp,
                                           const DevImage *__restrict__ images, const DevPatch *__restrict__ patches, int K,
                                           SrcImg *__restrict__ srcimg, Comp *__restrict__ comps) {
    // (called by whole wavefronts: the components on lanes < 14 K, the brightness moments by brightness_moments_wave -- twenty
    // lanes evaluate the twenty exponentials side by side instead of lane 63 one after the other, same bits; the serial version
    // was the wavefront's critical path: prep_kernel 0.18 ms for the 178 636 visits of config 5)
    const double *vs = vp + (size_t)s * CEL_P;
    const DevPatch &p = patches[sn];
    const int b = images[n].band - 1;
    prep_visit_values<false>(c, vs, p, b, K, srcimg + sn, comps + (size_t)sn * (14 * K));
    brightness_moments_wave(c, vs, p, b, srcimg + sn);
}

__global__ void __launch_bounds__(64)
prep_kernel(const double *__restrict__ vp, const DevImage *__restrict__ images,
            const DevPatch *__restrict__ patches, const int32_t *__restrict__ vis_src,
            const int32_t *__restrict__ vis_img, int N, int K,
            SrcImg *__restrict__ srcimg, Comp *__restrict__ comps,
            const int32_t *__restrict__ targets, const int32_t *__restrict__ vis_off, int M, int dense,
            const int32_t *__restrict__ live, const int32_t *__restrict__ mark, int32_t stamp) {
    // one workgroup per visit = (source, image) pair with a non-empty patch; the tables are indexed by visit (sn).
    // targets == nullptr: every visit of the context (neighbours are about to be rendered).  Otherwise one workgroup
    // per candidate visit k = ti * M + j of the batch's targets only (neighbours frozen).
    int s, n, sn;
    if (targets) {
        const int k = blockIdx.x, ti = k / M, j = k - ti * M;
        if (live && ti >= *live) return;
        s = targets[ti];
        n = j;
        if (!dense) {
            const int vo = vis_off[s];
            if (j >= vis_off[s + 1] - vo) return;
            n = vis_img[vo + j];
        }
        sn = dense ? s * N + n : vis_off[s] + j;
        const DevPatch &q = patches[sn];
        if (q.H2 * q.W2 <= 0) return;
    } else {
        s = vis_src[blockIdx.x]; n = vis_img[blockIdx.x];
        sn = blockIdx.x;
        // mark (optional): only the sources this batch reads -- its targets and their neighbours (setup_thread)
        if (mark && mark[s] != stamp) return;
    }
    prep_visit(threadIdx.x, s, n, sn, vp, images, patches, K, srcimg, comps);
}

// per-source shape derivatives and the finiteness flag of its parameters
__device__ __forceinline__ void source_geo_values(const double *vs, SrcGeo *out) {
    double x11, x12, x22;
    bvn_cov(vs[3], vs[4], vs[5], x11, x12, x22);
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
        *out = g;
}
__device__ inline void source_geo(const double *vp, int s, SrcGeo *geo) { source_geo_values(vp + (size_t)s * CEL_P, geo + s); }

// ---------------------------------------------------------------------------------------------
// Parameter-independent terms, once per context.
// iota_kernel: log(iota) per image row.
// patch_lgamma_kernel: sum of lgamma(pixel + 1) over the visited pixels of every (source, image) patch
// (elbo_objective.jl:391 subtracts it pixel by pixel; it depends on no parameter, so the pixel kernel neither reads an
// 8-byte plane for it nor carries it through the component loop: the lift subtracts the patch's constant).  One
// workgroup per visit, fixed-order reduction.
// ---------------------------------------------------------------------------------------------
__global__ void iota_kernel(const float *__restrict__ iota, int H, double *__restrict__ log_iota) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    // log(iota) is a Float32 log in the reference (iota::Float32, elbo_objective.jl:292)
    if (i < H) log_iota[i] = (double)(float)log((double)iota[i]);
}
__global__ void __launch_bounds__(256)
patch_lgamma_kernel(const DevImage *__restrict__ images, const DevPatch *__restrict__ patches,
                    const int32_t *__restrict__ vis_img, const uint8_t *__restrict__ bitmaps, double *__restrict__ out) {
    __shared__ double s_part[256];
    const int v = blockIdx.x;
    const DevPatch &P = patches[v];
    const DevImage &img = images[vis_img[v]];
    const int npx = P.H2 * P.W2;
    double s = 0.0;
    for (int idx = threadIdx.x; idx < npx; idx += 256) {
        const int w2 = idx / P.H2, h2 = idx - w2 * P.H2;
        const float x = img.pixels[(size_t)(P.off_h + h2) + (size_t)img.H * (P.off_w + w2)];
        bool valid = !isnan(x);                              // elbo_objective.jl:459
        if (valid && P.bitmap_off >= 0) valid = bitmaps[P.bitmap_off + h2 + (int64_t)P.H2 * w2] != 0;   // :445
        if (valid) s += lgamma((double)x + 1.0);
    }
    s_part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) s_part[threadIdx.x] += s_part[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[v] = s_part[0];
}

// ---------------------------------------------------------------------------------------------
// spline_prefilter_kernel: the ImagePatch constructor's arithmetic for every PSF stamp of a context
// (imaged_sources.jl:97-107: max(., 0), + 1e-6, normalise, softpluslike, then the prefilter of
// interpolate(., BSpline(Cubic(Line())), OnGrid()): a (1, 4, 1) / 6 tridiagonal solve along each axis on a padded 53 x 53
// grid) -- what celeste_spline_prefilter does for one stamp on the host.  Production Celeste has one stamp per (source,
// image) patch (SDSSPSFMap evaluated at every source): 8765 on the bench field, 55 us each on a host core = 0.48 s of
// celeste_ctx_create against a 0.06 s joint inference of the whole field.  One 64-thread workgroup per stamp; the same
// operations in the same order as the host function (no contraction; the sum of the stamp in index order), so the
// coefficients are the host's wherever the device's log agrees with libm's to the last bit, and within an ulp elsewhere.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void prefilter_line_dev(const double *__restrict__ d, int dstride, double *__restrict__ c, int cstride,
                                                   const double *__restrict__ cp) {
#pragma clang fp contract(off)
    constexpr int n = CEL_STAMP, m = CEL_STAMP - 2;
    const double x1 = d[0], xn = d[(n - 1) * dstride];
    // forward sweep: dp[q] is kept in its final place c[2 + q]
    double prev = 0.0;
    for (int q = 0; q < m; ++q) {
        double rhs = 6.0 * d[(q + 1) * dstride];
        if (q == 0) rhs -= x1;
        if (q == m - 1) rhs -= xn;
        const double lower = (q == 0) ? 0.0 : 1.0;
        const double denom = 4.0 - lower * (q ? cp[q - 1] : 0.0);
        prev = (rhs - lower * prev) / denom;
        c[(2 + q) * cstride] = prev;
    }
    double nxt = prev;                                    // x[2 + m - 1] = dp[m - 1]
    for (int q = m - 2; q >= 0; --q) {
        nxt = c[(2 + q) * cstride] - cp[q] * nxt;
        c[(2 + q) * cstride] = nxt;
    }
    c[1 * cstride] = x1;
    c[n * cstride] = xn;
    c[0] = 2.0 * x1 - c[2 * cstride];
    c[(n + 1) * cstride] = 2.0 * xn - c[(n - 1) * cstride];
}
__global__ void __launch_bounds__(64)
spline_prefilter_kernel(const double *__restrict__ stamps, double *__restrict__ coefs, float *__restrict__ coefs_f) {
#pragma clang fp contract(off)
    constexpr int n = CEL_STAMP, m = CEL_COEF;
    __shared__ double g[n * n];
    __shared__ double tmp[m * n];
    __shared__ double cp[n];
    __shared__ double s_sum;
    const int t = threadIdx.x;
    const double *__restrict__ s = stamps + (size_t)blockIdx.x * (n * n);
    double *__restrict__ out = coefs + (size_t)blockIdx.x * (m * m);
    for (int k = t; k < n * n; k += 64) g[k] = fmax(s[k], 0.0) + 1e-6;
    if (t == 1) {           // the Thomas factors of the (1, 4, 1) system: the same for every line
        double prev = 0.0;
        for (int q = 0; q < n - 2; ++q) { const double denom = 4.0 - (q ? 1.0 : 0.0) * prev; prev = 1.0 / denom; cp[q] = prev; }
    }
    __syncthreads();
    if (t == 0) {           // the host's sum, in its order
        double sum = 0.0;
        for (int k = 0; k < n * n; ++k) sum += g[k];
        s_sum = sum;
    }
    __syncthreads();
    const double sum = s_sum;
    for (int k = t; k < n * n; k += 64) {
        const double x = g[k] / sum;
        g[k] = (1000 * x > 1) ? 1000 * x - 1 : log(1000 * x);   // softpluslike (fsm_util.jl:221)
    }
    __syncthreads();
    if (t < n) prefilter_line_dev(g + n * t, 1, tmp + m * t, 1, cp);
    __syncthreads();
    if (t < m) prefilter_line_dev(tmp + t, m, out + t, m, cp);
    __syncthreads();
    __threadfence_block();
    float *__restrict__ outf = coefs_f + (size_t)blockIdx.x * (m * m);
    for (int k = t; k < m * m; k += 64) outf[k] = (float)out[k];
}

// ---------------------------------------------------------------------------------------------
// pixel_kernel
// ---------------------------------------------------------------------------------------------
template <typename S>
__device__ inline void bspline_w(S f, S w[4]) {
    const S o = (S)1.0 - f;
    w[0] = o * o * o * (S)(1.0 / 6); w[1] = (S)(2.0 / 3) - f * f + f * f * f * (S)0.5;
    w[2] = (S)(2.0 / 3) - o * o + o * o * o * (S)0.5; w[3] = f * f * f * (S)(1.0 / 6);
}
template <typename S>
__device__ inline void bspline_dw(S f, S dw[4], S ddw[4]) {
    const S o = (S)1.0 - f;
    dw[0] = (S)-0.5 * o * o; dw[1] = (S)-2 * f + (S)1.5 * f * f; dw[2] = (S)2 * o - (S)1.5 * o * o; dw[3] = (S)0.5 * f * f;
    ddw[0] = o; ddw[1] = (S)-2 + (S)3 * f; ddw[2] = (S)-2 + (S)3 * o; ddw[3] = f;
}
// exp / log in the arithmetic type of the per-pixel terms
__device__ __forceinline__ double exp_s(double x) { return exp(x); }
__device__ __forceinline__ float exp_s(float x) { return __expf(x); }
__device__ __forceinline__ double log_s(double x) { return log(x); }
__device__ __forceinline__ float log_s(float x) { return __logf(x); }

// exp(x) for x <= 0 in fp64.  x = (64 m + j) ln2/64 + r with |r| <= ln2/128, so
// exp(x) = 2^m * 2^(j/64) * exp(r): a 64-entry table in LDS (filled by exp_table_init), a degree-4
// Taylor polynomial (truncation 3.9e-14 relative; EXP_DEGREE 5 gives 3.5e-17 for one more FMA) and
// v_ldexp_f64.  16 VALU + 1 LDS instruction instead of ~31 for the library exp.  Inputs below -745 give
// exactly 0 like the reference's exp.
#ifndef EXP_TAB_LOG2
#define EXP_TAB_LOG2 6   // table of 2^(j / T), T = 2^EXP_TAB_LOG2 entries (the 64-double LDS array holds 64 / T copies)
#endif
#ifndef EXP_DEGREE
#define EXP_DEGREE (EXP_TAB_LOG2 == 6 ? 4 : EXP_TAB_LOG2 == 5 ? 5 : 6)   // Taylor degree of exp(r), |r| <= ln2 / 2T: truncation <= 4.5e-14 relative
#endif
#define EXP_T (1 << EXP_TAB_LOG2)
#define EXP_INV_STEP (92.33248261689366 / (64 / EXP_T))              // T / ln 2
#define EXP_STEP_HI (0.010830424696450791 * (64 / EXP_T))            // ln2 / T, high part (n * hi exact for |n| < 2^17)
#define EXP_STEP_LO (2.0164562921995537e-13 * (64 / EXP_T))          // minus its low part
#define EXP_STEP_FULL (0.010830424696249145 * (64 / EXP_T))           // ln2 / T rounded to double (0x1.62e42fefa39efp-7 for T = 64)
__device__ double g_exp2_table[64];  // 2^((j mod T) / T), filled once per context by exp_table_kernel
// (degree 4: the table holds 2^(j / T) / 24 and exp_poly returns 24 exp(r) -- see exp_poly)
#define EXP_TAB_SCALE (EXP_DEGREE == 4 ? 1.0 / 24.0 : 1.0)
__global__ void exp_table_kernel() { g_exp2_table[threadIdx.x] = exp2((double)(threadIdx.x & (EXP_T - 1)) * (1.0 / EXP_T)) * EXP_TAB_SCALE; }
__device__ __forceinline__ void exp_table_init(double *tab) {
    if (threadIdx.x < 64) tab[threadIdx.x] = g_exp2_table[threadIdx.x];  // one coalesced 512-byte read
    __syncthreads();
}
// exp(r) -- as a polynomial in r of degree EXP_DEGREE (Horner).  Degree 4 (the build's default): 24 exp(r) =
// (((r + 4) r + 12) r + 24) r + 24, the factor 1 / 24 sitting in the table -- one add with an inline constant and three FMAs
// with ONE non-inline constant each (a scalar register operand), where the monic form 1/24, 1/6, 1/2, 1, 1 needs its first two
// constants in one instruction, i.e. a v_mov_b64 into a vector register per evaluation: 5 -> 4 VALU instructions.
__device__ __forceinline__ double exp_poly(double r) {
    double p;
#if EXP_DEGREE == 4
    p = r + 4.0;
    p = __builtin_fma(p, r, 12.0);
    p = __builtin_fma(p, r, 24.0);
    return __builtin_fma(p, r, 24.0);
#endif
#if EXP_DEGREE >= 6
    p = 1.3888888888888889e-03;                      // 1/6!
    p = __builtin_fma(p, r, 8.333333333333333e-03);  // 1/5!
    p = __builtin_fma(p, r, 4.1666666666666664e-02); // 1/4!
#elif EXP_DEGREE == 5
    p = 8.333333333333333e-03;
    p = __builtin_fma(p, r, 4.1666666666666664e-02);
#else
    p = 4.1666666666666664e-02;                      // truncation |r|^5 / 120 <= 3.9e-14 for |r| <= ln2 / 128
#endif
    p = __builtin_fma(p, r, 1.6666666666666666e-01); // 1/3!
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_fma(p, r, 1.0);
}
__device__ __forceinline__ double exp_nonpos(double x, const double *tab) {
    // no clamp is needed: for x << -745 the integer part saturates and v_ldexp_f64 returns 0
    const double n = rint(x * EXP_INV_STEP);
    double r = __builtin_fma(n, -EXP_STEP_HI, x);
    r = __builtin_fma(n, EXP_STEP_LO, r);
    const int ni = (int)n;
    const double tj = tab[ni & (EXP_T - 1)];
    return ldexp(exp_poly(r) * tj, ni >> EXP_TAB_LOG2);
}

// log(x) for the per-pixel term (x = E_G > 0, normal): x = 2^e m with m in [sqrt(1/2), sqrt 2), s = (m - 1) / (m + 1),
// log m = 2 s (1 + s^2/3 + s^4/5 + ... + s^20/21) (truncation 1e-17 relative), e ln2 in a high part whose product with e is
// exact and a low part.  <= 2 ulp against a 40-digit reference over 1e-5 .. 1e6 and around 1 (the micro-parity tests
// hold the kernel to a libm log at 1e-12 on the pixel term); 32 instructions where the library's is ~80.
// Non-positive, subnormal or NaN arguments (never reached on valid inputs) give NaN, +inf gives +inf.
__device__ __forceinline__ double rcp_pos(double b) {      // 1 / b, b > 0 normal: hardware estimate + two Newton steps (<= 1 ulp)
    double r = __builtin_amdgcn_rcp(b);
    r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
    return __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
}
__device__ __forceinline__ double log_pos(double x) {
    const bool ok = x > 2.2250738585072014e-308 && x < INFINITY;   // else NaN (x <= 0, subnormal, NaN) or +inf: no library path
    const double bad = x == INFINITY ? x : __builtin_nan("");
    double m = __builtin_amdgcn_frexp_mant(x);             // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(x);
    const bool low = m < 0.7071067811865476;
    m = low ? m + m : m; e = low ? e - 1 : e;
    const double a = m - 1.0, b = m + 1.0, r = rcp_pos(b);
    double sq = a * r;
    sq = __builtin_fma(__builtin_fma(-b, sq, a), r, sq);   // (m - 1) / (m + 1) to half an ulp
    const double z = sq * sq;
    double p = 1.0 / 21.0;
    p = __builtin_fma(p, z, 1.0 / 19.0); p = __builtin_fma(p, z, 1.0 / 17.0); p = __builtin_fma(p, z, 1.0 / 15.0);
    p = __builtin_fma(p, z, 1.0 / 13.0); p = __builtin_fma(p, z, 1.0 / 11.0); p = __builtin_fma(p, z, 1.0 / 9.0);
    p = __builtin_fma(p, z, 1.0 / 7.0); p = __builtin_fma(p, z, 1.0 / 5.0); p = __builtin_fma(p, z, 1.0 / 3.0);
    const double t = sq * z * p, ef = (double)e;
    const double lg = __builtin_fma(ef, 0x1.62e42fee00000p-1, (sq + sq) + __builtin_fma(ef, 0x1.a39ef35793c76p-33, t + t));
    return ok ? lg : bad;
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

// value of the galaxy density sum_c w0_c exp(-0.5 d' P d) (populate_gal_fsm!, inactive branch)
// (dx, dy) = pixel - m_pos
__device__ inline double galaxy_value(const Comp *tc, int NC, double dx, double dy, const double *etab) {
    double v = 0;
    // the 8 (de Vaucouleurs) / 6 (exponential) prototypes of a run share a PSF component, i.e. the offset xiBar_k (records are
    // ordered type, PSF component, prototype): d and -d / 2 once per run, as in the derivative loops
    const int n_dev = 8 * (NC / 14);
    for (int c0 = 0; c0 < NC; c0 += (c0 < n_dev ? 8 : 6)) {
        const int len = c0 < n_dev ? 8 : 6;
        const double d1 = dx - tc[c0].xi1, d2 = dy - tc[c0].xi2;
        // value only: -d' P d / 2 = p11 (-d1^2 / 2) + p12 (-d1 d2) + p22 (-d2^2 / 2) -- the three products of the offset once per
        // run, three instructions per component where P d and d' (P d) take six (the derivative loops need P d itself)
        const double s11 = -0.5 * (d1 * d1), s12 = -(d1 * d2), s22 = -0.5 * (d2 * d2);
        for (int c = c0; c < c0 + len; ++c) {
            const Comp k = tc[c];
            v = __builtin_fma(k.w0, exp_nonpos(__builtin_fma(k.p11, s11, __builtin_fma(k.p12, s12, k.p22 * s22)), etab), v);
        }
    }
    return v;
}

// The same two densities in single precision (CELESTE_FLAG_FP32: the neighbours' light is rendered in the arithmetic of
// the per-pixel terms).
__device__ inline float star_value_f(const double *__restrict__ coef, float xh, float xw) {
    int ix = (int)floorf(xh); ix = ix < 1 ? 1 : (ix > 50 ? 50 : ix);
    int iy = (int)floorf(xw); iy = iy < 1 ? 1 : (iy > 50 ? 50 : iy);
    float wx[4], wy[4];
    bspline_w(xh - (float)ix, wx); bspline_w(xw - (float)iy, wy);
    const double *c0 = coef + (ix - 1) + CEL_COEF * (iy - 1);
    float y = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const double *cb = c0 + CEL_COEF * b;
        const float r = (float)cb[0] * wx[0] + (float)cb[1] * wx[1] + (float)cb[2] * wx[2] + (float)cb[3] * wx[3];
        y += r * wy[b];
    }
    return y < 0 ? 1e-3f * __expf(y) : 1e-3f * (y + 1.0f);
}
// two components per instruction (v_pk_fma_f32): tcf holds, per PAIR of components, 6 slots of two floats --
// p11, p12, p22, xi1, xi2, w0 of components c, c + 1 in the two halves (NC = 14 psf_K is even)
typedef float vf2 __attribute__((ext_vector_type(2)));
__device__ inline float galaxy_value_f(const float *tcf, int NC, float dx, float dy) {
    const vf2 *tp = reinterpret_cast<const vf2 *>(tcf);
    const vf2 dxx = (vf2)(dx), dyy = (vf2)(dy);
    vf2 v = (vf2)(0.0f);
    for (int c = 0; c < NC; c += 2) {
        const vf2 *k = tp + 3 * c;
        const vf2 p11 = k[0], p12 = k[1], p22 = k[2], xi1 = k[3], xi2 = k[4], w0 = k[5];
        const vf2 d1 = dxx - xi1, d2 = dyy - xi2;
        const vf2 u = p11 * d1 + p12 * d2, vv = p12 * d1 + p22 * d2;
        const vf2 q = -0.5f * (d1 * u + d2 * vv);
        const vf2 e = {__expf(q.x), __expf(q.y)};
        v += w0 * e;
    }
    return v.x + v.y;
}

// idx / d and idx mod d for 0 <= idx < 2^22 and 1 <= d: one float multiply and a correction instead of the ~35 instructions of
// a 32-bit integer division by a run-time divisor (rd = 1.0f / d, formed once per patch).  Exact: (float)idx is exact, the
// product is within one of the true quotient, and the remainder test puts it right.
__device__ __forceinline__ void divmod_small(int idx, int d, float rd, int &q, int &r) {
    q = (int)((float)idx * rd);
    r = idx - q * d;
    if (r < 0) { --q; r += d; }
    else if (r >= d) { ++q; r -= d; }
}

// ---- cross-lane helpers (wave64) ----------------------------------------------------------------
__device__ inline double wave_sum(double x) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

// Loads / stores of data that another workgroup of the SAME launch wrote or will read (optim_fused_kernel hands targets
// from workgroup to workgroup): COH = true makes them agent-scope relaxed atomics, i.e. `global_load / global_store ...
// sc1` -- write-through stores and L1-bypassing loads, so that no release / acquire fence is needed around them (guide
// section 6, Guideline 16, form R1).  COH = false: plain accesses (data crosses kernel boundaries only).
template <bool COH>
__device__ __forceinline__ double ldc(const double *p) {
    if constexpr (COH)
        return __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED,
                                                            __HIP_MEMORY_SCOPE_AGENT));
    else return *p;
}
template <bool COH>
__device__ __forceinline__ void stc(double *p, double v) {
    if constexpr (COH)
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
template <bool COH>
__device__ __forceinline__ int ldc(const int *p) {
    if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
template <bool COH>
__device__ __forceinline__ void stc(int *p, int v) {
    if constexpr (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// ---------------------------------------------------------------------------------------------
// mark_kernel / value_kernel: value-only light of every source that is some target's neighbour
// (is_active_source == false branch of add_pixel_term!, elbo_objective.jl:254-257), rendered once
// per batch on the source's own patch so that the pixel kernel gathers two doubles per covering
// neighbour instead of re-evaluating 14 psf_K exponentials per (pixel, neighbour) pair.
// ---------------------------------------------------------------------------------------------
// items[ti * M + j] = {visit id, image} of the j-th visit of target ti (-1: none), so that the pixel kernels find their
// (target, image) with two independent loads instead of a chain through the visit lists
// (filled by setup_thread)

// Table entry (visit) of (source, image) for the kernels off the hot path: s N + n when every source is listed in
// every image, else a search of the source's visit list (-1: the source has no patch in that image)
__device__ __forceinline__ int table_entry(const int32_t *__restrict__ vis_off, const int32_t *__restrict__ vis_img,
                                           int dense, int N, int s, int n) {
    if (dense) return s * N + n;
    for (int v = vis_off[s]; v < vis_off[s + 1]; ++v) if (vis_img[v] == n) return v;
    return -1;
}

// One launch for the per-batch bookkeeping: SrcGeo of every source, the visit items of the batch (unless every
// source is listed in every image), the target flags (is_target == nullptr: keep an earlier rendering)
__device__ inline void source_geo(const double *vp, int s, SrcGeo *geo);
// The target marks are batch stamps (is_target[s] == stamp: s is a target of THIS launch): no clearing pass between
// launches.
__device__ inline void setup_thread(int k, const double *__restrict__ vp, int S, SrcGeo *__restrict__ geo,
                                    const int32_t *__restrict__ targets, int n_targets, const int32_t *__restrict__ vis_off,
                                    const int32_t *__restrict__ vis_img, int M, int2 *__restrict__ items,
                                    int32_t *__restrict__ is_target, int32_t stamp, int32_t *__restrict__ prep_mark,
                                    const int64_t *__restrict__ nbr_off, const int32_t *__restrict__ nbr_idx,
                                    const int32_t *__restrict__ live = nullptr) {
    // live (optional): the number of leading entries of `targets` that are valid -- n_targets is then the host's upper
    // bound (the chained optimiser runs ahead of the device) and the entries past *live are stale or never written
    if (live) n_targets = min(n_targets, *live);
    // S < 0: the neighbours are frozen (an optimiser iteration) -- only the targets have moved since the batch's SrcGeo
    // table was made, and the neighbours' entries (their finiteness flags) keep describing the parameters they were
    // rendered with
    if (S < 0) { if (k < n_targets) source_geo(vp, targets[k], geo); }
    else if (k < S) source_geo(vp, k, geo);
    if (prep_mark && k < n_targets) {   // the sources whose per-image tables this batch reads
        const int t = targets[k];
        prep_mark[t] = stamp;
        for (int64_t q = nbr_off[t]; q < nbr_off[t + 1]; ++q) prep_mark[nbr_idx[q]] = stamp;
    }
    if (items && k < n_targets * M) {
        const int ti = k / M, j = k - ti * M;
        const int t = targets[ti];
        const int vo = vis_off[t];
        items[k] = j < vis_off[t + 1] - vo ? make_int2(vo + j, vis_img[vo + j]) : make_int2(-1, -1);
    }
    if (is_target && k < n_targets) is_target[targets[k]] = stamp;
}
__global__ void setup_kernel(const double *__restrict__ vp, int S, SrcGeo *__restrict__ geo,
                             const int32_t *__restrict__ targets, int n_targets, const int32_t *__restrict__ vis_off,
                             const int32_t *__restrict__ vis_img, int M, int2 *__restrict__ items,
                             int32_t *__restrict__ is_target, int32_t stamp, int32_t *__restrict__ prep_mark,
                             const int64_t *__restrict__ nbr_off, const int32_t *__restrict__ nbr_idx,
                             const int32_t *__restrict__ live) {
    setup_thread(blockIdx.x * blockDim.x + threadIdx.x, vp, S, geo, targets, n_targets, vis_off, vis_img, M, items, is_target,
                 stamp, prep_mark, nbr_off, nbr_idx, live);
}

// ---------------------------------------------------------------------------------------------
// Work list of pixel_kernel.  A batch has n_targets x M candidate visits k = ti * M + j (j-th image of target ti) and
// each is cut into ceil(npx / chunk_px) chunks, 0 for a target that appears in fewer than M images.  Launching the
// full n_targets x M x CH grid leaves most workgroups with nothing to do (patches of 200 .. 2600 pixels; 5 .. 20
// images per source in a many-field problem), and an idle workgroup still holds a wave slot with its full register
// allocation for three dependent loads before it can exit.  The list enumerates exactly the chunks that exist,
// longest first: class 0 = full chunks (chunk_px / 64 iterations of the pixel loop), class c = the last, partial
// chunk of a patch that needs c iterations fewer -- so the kernel's tail is made of its shortest workgroups.
// Three small kernels: per-block counts, one exclusive scan over [class][block], fill.  The order is deterministic.
// ---------------------------------------------------------------------------------------------
#define WORK_NT 256
#define WORK_CLASSES 4
// A work item is a GROUP of up to G consecutive chunks of one patch, handled by one workgroup: the workgroup's fixed
// cost (a chain of four dependent loads, staging of the component records, ~4 us of its wave slot) is paid once per
// group, while every chunk still produces its own record -- so the results do not depend on G, which the host picks per
// launch (large sweeps: 4; small, latency-bound batches: 1).
// Groups of candidate visit k: n_full of class 0 (first chunks 0, G, 2 G, ...: G full chunks each) and, if
// last_class > 0, one last group of that class (first chunk n_full * G) which is last_class pixel-loop iterations short.
__device__ inline void visit_chunks(int k, const int32_t *__restrict__ targets, const DevPatch *__restrict__ patches,
                                    const int32_t *__restrict__ vis_off, const int32_t *__restrict__ vis_img, int N, int M,
                                    int chunk_px, int G, bool dense, int &n_full, int &last_class, int *n_chunks = nullptr) {
    n_full = 0; last_class = 0;
    if (n_chunks) *n_chunks = 0;
    const int ti = k / M, j = k - ti * M;
    const int t = targets[ti];
    int n = j;
    if (!dense) {
        const int vo = vis_off[t];
        if (j >= vis_off[t + 1] - vo) return;
        n = vis_img[vo + j];
    }
    const DevPatch &P = patches[dense ? t * N + n : vis_off[t] + j];
    const int npx = P.H2 * P.W2;
    if (npx <= 0) return;
    if (n_chunks) *n_chunks = (npx + chunk_px - 1) / chunk_px;
    const int gpx = chunk_px * G;                        // pixels of a full group
    const int ngr = (npx + gpx - 1) / gpx;
    const int last_px = npx - (ngr - 1) * gpx;           // 1 .. gpx pixels in the last group
    const int ipc = chunk_px >> 6;                       // iterations of a full chunk
    const int last_iters = (last_px / chunk_px) * ipc + ((last_px % chunk_px + 63) >> 6);
    const int shortfall = G * ipc - last_iters;          // 0 .. G ipc - 1 iterations short of a full group
    // WORK_CLASSES - 1 length classes for the last groups (quantised shortfall): the order only has to be roughly
    // longest first, and every class costs the list kernels a ballot pass
    last_class = shortfall == 0 ? 0 : 1 + ((shortfall - 1) * (WORK_CLASSES - 1)) / (G * ipc - 1);
    n_full = last_class == 0 ? ngr : ngr - 1;
}

// rank of this thread among the threads of its block for which pred holds (exclusive), and the block total
__device__ inline int block_rank(bool pred, int *s_wave /* WORK_NT / 64 */, int &total) {
    const unsigned long long m = __ballot(pred);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = __popcll(m & ((1ull << lane) - 1ull));
    __syncthreads();
    if (lane == 0) s_wave[wv] = __popcll(m);
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < WORK_NT / 64; ++w) { const int c = s_wave[w]; off += w < wv ? c : 0; tot += c; }
    total = tot;
    return off + r;
}
// exclusive prefix sum of v over the threads of the block, and the block total
__device__ inline int block_prefix(int v, int *s_wave /* WORK_NT / 64 */, int &total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(x, d); if (lane >= d) x += y; }
    __syncthreads();
    if (lane == 63) s_wave[wv] = x;
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < WORK_NT / 64; ++w) { const int c = s_wave[w]; off += w < wv ? c : 0; tot += c; }
    total = tot;
    return off + x - v;
}

__global__ void __launch_bounds__(WORK_NT)
work_count_kernel(const int32_t *__restrict__ targets, int n_visits, const DevPatch *__restrict__ patches,
                  const int32_t *__restrict__ vis_off, const int32_t *__restrict__ vis_img, int N, int M,
                  int chunk_px, int G, int dense, int32_t *__restrict__ blk_cnt /* [WORK_CLASSES][gridDim.x] */,
                  const int32_t *__restrict__ live) {
    __shared__ int s_wave[WORK_NT / 64];
    const int k = blockIdx.x * WORK_NT + threadIdx.x;
    if (live) n_visits = min(n_visits, *live * M);   // the host's target count is an upper bound (device-resident loops)
    int n_full = 0, lc = 0, nch = 0;
    if (k < n_visits) visit_chunks(k, targets, patches, vis_off, vis_img, N, M, chunk_px, G, dense != 0, n_full, lc, &nch);
    int tot;
    block_prefix(n_full, s_wave, tot);
    if (threadIdx.x == 0) blk_cnt[blockIdx.x] = tot;
    for (int c = 1; c < WORK_CLASSES; ++c) {
        block_rank(lc == c, s_wave, tot);
        if (threadIdx.x == 0) blk_cnt[(size_t)c * gridDim.x + blockIdx.x] = tot;
    }
    // row WORK_CLASSES: chunks per block, for the record offsets of the visits (rec_off: a visit's chunk records are
    // consecutive, visits in order -- the record buffer holds exactly the chunks that exist)
    block_prefix(nch, s_wave, tot);
    if (threadIdx.x == 0) blk_cnt[(size_t)WORK_CLASSES * gridDim.x + blockIdx.x] = tot;
}

// exclusive scan of cnt[0 .. n) in place, total -> *total (one workgroup)
// total_at: *total = the exclusive prefix at that index (= the sum of the rows before it); total_at == n: the grand total
__global__ void __launch_bounds__(1024) work_scan_kernel(int32_t *__restrict__ cnt, int n, int32_t *__restrict__ total,
                                                         int total_at) {
    __shared__ int s_part[16];
    __shared__ int s_run;
    if (threadIdx.x == 0) s_run = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int base = 0; base < n; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < n ? cnt[i] : 0;
        int x = v;   // inclusive scan within the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(x, d); if (lane >= d) x += y; }
        if (lane == 63) s_part[wv] = x;
        __syncthreads();
        int off = s_run;
        for (int w = 0; w < wv; ++w) off += s_part[w];
        if (i < n) cnt[i] = off + x - v;
        if (i == total_at) *total = off + x - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_run = off + x;
        __syncthreads();
    }
    if (threadIdx.x == 0 && total_at >= n) *total = s_run;
}

__global__ void __launch_bounds__(WORK_NT)
work_fill_kernel(const int32_t *__restrict__ targets, int n_visits, const DevPatch *__restrict__ patches,
                 const int32_t *__restrict__ vis_off, const int32_t *__restrict__ vis_img, int N, int M, int CH,
                 int chunk_px, int G, int dense, const int32_t *__restrict__ blk_base, int32_t *__restrict__ work,
                 const int32_t *__restrict__ live, int32_t *__restrict__ rec_off) {
    __shared__ int s_wave[WORK_NT / 64];
    const int k = blockIdx.x * WORK_NT + threadIdx.x;
    if (live) n_visits = min(n_visits, *live * M);
    int n_full = 0, lc = 0, nch = 0;
    if (k < n_visits) visit_chunks(k, targets, patches, vis_off, vis_img, N, M, chunk_px, G, dense != 0, n_full, lc, &nch);
    int tot;
    {   // first chunk record of visit k: chunks of the blocks before (scanned row WORK_CLASSES, minus the rows before it) + in-block prefix
        const int base = blk_base[(size_t)WORK_CLASSES * gridDim.x + blockIdx.x] - blk_base[(size_t)WORK_CLASSES * gridDim.x];
        const int pre = block_prefix(nch, s_wave, tot);
        if (k < n_visits) rec_off[k] = base + pre;
    }
    const int p0 = blk_base[blockIdx.x] + block_prefix(n_full, s_wave, tot);
    for (int g = 0; g < n_full; ++g) work[p0 + g] = k * CH + g * G;     // record index of the group's first chunk
    for (int c = 1; c < WORK_CLASSES; ++c) {
        const int r = block_rank(lc == c, s_wave, tot);
        if (lc == c) work[blk_base[(size_t)c * gridDim.x + blockIdx.x] + r] = k * CH + n_full * G;
    }
}

// The record range of target ti of a batch and the description of each of its chunk records (what a workgroup of the fused
// kernels needs to start on a record, in one 32-byte read): fused_setup_kernel's per-target work; block 0 of
// setup_worklist_kernel does it itself for small Hessian-mode batches on contexts whose sources are listed in every image
// (one launch less in front of eval_fused_kernel: a one-target call is a chain of three kernels instead of four).
// Returns the number of records; first = the first record (-1: none).
__device__ __forceinline__ int describe_target_records(int ti, int t, const DevPatch *__restrict__ patches, const int2 *__restrict__ items,
                                                       int N, int M, int chunk_px, const int32_t *rec_off, int4 *__restrict__ chunk_desc,
                                                       int &first) {
    int n_rec = 0;
    first = -1;
    for (int j = 0; j < M; ++j) {
        const int tn = ti * M + j;
        int v = t * N + j, n = j;                // items == nullptr: every source is listed in all M = N images
        if (items) { v = items[tn].x; n = items[tn].y; }
        if (v < 0) continue;
        const DevPatch &P = patches[v];
        const int npx = P.H2 * P.W2;
        if (npx <= 0) continue;
        const int nch = (npx + chunk_px - 1) / chunk_px, r0 = rec_off[tn];
        if (first < 0) first = r0;
        for (int ch = 0; ch < nch; ++ch) {
            chunk_desc[2 * (r0 + ch)] = make_int4(ti, j, ch, t);
            chunk_desc[2 * (r0 + ch) + 1] = make_int4(v, n, 0, 0);
        }
        n_rec += nch;
    }
    return n_rec;
}

// setup_kernel and the three work-list kernels in ONE launch, for batches of up to WORK1_MAX_VISITS candidate visits
// (a rank's shard of a field, an optimiser iteration, a Cyclades layer): block 0 (1024 threads, every thread a
// contiguous run of visits) counts, scans and fills the list; blocks 1.. are setup_kernel.  Four launches and their
// drain / fill gaps become one: 44 -> 35 us in front of the pixel kernel for a 250-target shard.  Larger batches keep
// the parallel three-kernel path (for the 2000-target sweep it is 13 us faster than one block looping).  Same list, same
// order.
#define WORK1_NT 1024
#define WORK1_SETUP 64        // setup_thread items per block of the fused launch (blocks 1 ..)
#define WORK1_MAX_VISITS 4096
#define WORK1_PREP_ALL_MAX 16384   // visits of a context whose tables the fused launch fills wholesale (neighbours rendered)
#define WORK1_PREP_ALL_MIN_TARGETS 64
#define WORK1_PREP_ALL_TINY 128     // contexts of up to this many visits: wholesale for every batch size
__global__ void __launch_bounds__(WORK1_NT)
setup_worklist_kernel(const double *__restrict__ vp, int S, SrcGeo *__restrict__ geo, const int32_t *__restrict__ targets,
                      int n_targets, const int32_t *__restrict__ vis_off, const int32_t *__restrict__ vis_img, int M,
                      int2 *__restrict__ items, int32_t *__restrict__ is_target, int32_t stamp,
                      const DevPatch *__restrict__ patches, int N, int CH, int chunk_px, int G, int dense,
                      int32_t *__restrict__ work, int32_t *__restrict__ work_total, const int32_t *__restrict__ live,
                      int32_t *__restrict__ prep_mark, const int64_t *__restrict__ nbr_off,
                      const int32_t *__restrict__ nbr_idx, int32_t *rec_off,
                      int setup_blocks, const DevImage *__restrict__ images, int K, SrcImg *__restrict__ srcimg,
                      Comp *__restrict__ comps, int prep_all_V, const int32_t *__restrict__ vis_src,
                      int4 *__restrict__ fz_chunk_desc = nullptr, int2 *__restrict__ fz_tgt_rec = nullptr, int fz_chunk_px = 0) {
    // fz_*: (optional; dense contexts, no `live`) the record descriptions of eval_fused_kernel, by block 0 behind its list
    if ((int)blockIdx.x > setup_blocks) {
        const int k = ((int)blockIdx.x - setup_blocks - 1) * (WORK1_NT / 64) + (int)(threadIdx.x >> 6);
        if (prep_all_V > 0) {
            // neighbours about to be rendered, a context of few visits: the tables of EVERY visit are filled by this
            // launch (prep_kernel's all-visits mode without a launch of its own, and without waiting for the marks)
            if (k < prep_all_V) prep_visit(threadIdx.x & 63, vis_src[k], vis_img[k], k, vp, images, patches, K, srcimg, comps);
            return;
        }
        // frozen neighbours (an optimiser iteration): the tables of the targets' own visits are filled by this launch
        // too, one wavefront per candidate visit -- prep_kernel's targets mode without a launch of its own
        const int ti = k / M, j = k - ti * M;
        if (ti >= n_targets || (live && ti >= *live)) return;
        const int s = targets[ti];
        int n = j;
        if (!dense) {
            const int vo = vis_off[s];
            if (j >= vis_off[s + 1] - vo) return;
            n = vis_img[vo + j];
        }
        const int sn = dense ? s * N + n : vis_off[s] + j;
        const DevPatch &q = patches[sn];
        if (q.H2 * q.W2 <= 0) return;
        prep_visit(threadIdx.x & 63, s, n, sn, vp, images, patches, K, srcimg, comps);
        return;
    }
    if (blockIdx.x > 0) {
        // one wavefront of setup work per block: a thread reads 44 and writes 37 doubles of ITS source, every wave
        // instruction touches 64 cache lines -- 16 such waves on one CU queued behind each other on its memory
        // pipeline (26 us for 2000 sources in two 1024-thread blocks; spread over 32 CUs the same work takes 3 us)
        if (threadIdx.x < WORK1_SETUP)
            setup_thread((blockIdx.x - 1) * WORK1_SETUP + threadIdx.x, vp, S, geo, targets, n_targets, vis_off, vis_img, M,
                         items, is_target, stamp, prep_mark, nbr_off, nbr_idx, live);
        return;
    }
    __shared__ int s_part[WORK1_NT / 64];
    __shared__ int s_base[WORK_CLASSES + 2];
    int n_visits = n_targets * M;
    if (live) n_visits = min(n_visits, *live * M);
    const int per = (n_visits + WORK1_NT - 1) / WORK1_NT;
    const int k0 = threadIdx.x * per, k1 = min(n_visits, k0 + per);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    constexpr int NCL = WORK_CLASSES + 1;   // class WORK_CLASSES counts chunks: the record offsets of the visits (rec_off)
    int cnt[NCL];
#pragma unroll
    for (int c = 0; c < NCL; ++c) cnt[c] = 0;
    for (int k = k0; k < k1; ++k) {
        int n_full, lc, nch;
        visit_chunks(k, targets, patches, vis_off, vis_img, N, M, chunk_px, G, dense != 0, n_full, lc, &nch);
        cnt[0] += n_full;
#pragma unroll
        for (int c = 1; c < WORK_CLASSES; ++c) cnt[c] += lc == c;
        cnt[WORK_CLASSES] += nch;
    }
    // exclusive scan over (class, thread): class totals first, then the threads inside each class
    int off[NCL];
    if (threadIdx.x == 0) s_base[0] = 0;
#pragma unroll
    for (int c = 0; c < NCL; ++c) {
        int x = cnt[c];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(x, d); if (lane >= d) x += y; }
        __syncthreads();
        if (lane == 63) s_part[wv] = x;
        __syncthreads();
        int before = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < WORK1_NT / 64; ++w) { const int v = s_part[w]; before += w < wv ? v : 0; tot += v; }
        off[c] = before + x - cnt[c];
        if (threadIdx.x == 0) s_base[c + 1] = s_base[c] + tot;
    }
    __syncthreads();
    if (threadIdx.x == 0) *work_total = s_base[WORK_CLASSES];
#pragma unroll
    for (int c = 0; c < WORK_CLASSES; ++c) off[c] += s_base[c];
    for (int k = k0; k < k1; ++k) {
        int n_full, lc, nch;
        visit_chunks(k, targets, patches, vis_off, vis_img, N, M, chunk_px, G, dense != 0, n_full, lc, &nch);
        for (int g = 0; g < n_full; ++g) work[off[0]++] = k * CH + g * G;
#pragma unroll
        for (int c = 1; c < WORK_CLASSES; ++c) if (lc == c) work[off[c]++] = k * CH + n_full * G;
        rec_off[k] = off[WORK_CLASSES];      // (not offset by s_base: the chunk class starts at 0)
        off[WORK_CLASSES] += nch;
    }
    if (fz_tgt_rec) {
        __syncthreads();      // rec_off is complete (written by the threads of this block)
        for (int ti = threadIdx.x; ti < n_targets; ti += WORK1_NT) {
            int first;
            const int n_rec = describe_target_records(ti, targets[ti], patches, nullptr, N, M, fz_chunk_px, rec_off, fz_chunk_desc, first);
            fz_tgt_rec[ti] = make_int2(first < 0 ? 0 : first, n_rec);
        }
    }
}

// the light of one neighbour on pixels [p0, p1) of an overlap rectangle (h fastest, RH rows, corner (h_lo, w_lo) in 0-based
// image coordinates), into the neighbour's own patch buffer `out`: value_kernel's loop, one wavefront.  COH: the values
// are for workgroups of the same launch (write-through stores).
template <bool COH, typename R = double>
__device__ __forceinline__ void value_pixels(int lane, const DevPatch &P, const SrcImg &si, const Comp *__restrict__ tc, int NC,
                                             const double *__restrict__ coefs, const double *__restrict__ etab, int h_lo,
                                             int w_lo, int RH, int p0, int p1, double2 *__restrict__ out,
                                             const float *__restrict__ tcf = nullptr) {
    const double *__restrict__ coef = coefs + (size_t)(CELESTE_MUTANT == 3 ? 0 : P.stamp) * (CEL_COEF * CEL_COEF);
    const double sh0 = 26.0 - si.m1, sw0 = 26.0 - si.m2;
    const float rRH = 1.0f / (float)RH;
    for (int idx = p0 + lane; idx < p1; idx += 64) {
        int rw, rh;
        divmod_small(idx, RH, rRH, rw, rh);
        const int h0 = h_lo + rh, w0 = w_lo + rw;  // 0-based image coordinates
        const double hh = (double)(h0 + 1), ww = (double)(w0 + 1);
        double f0, f1;
        if constexpr (sizeof(R) == 8) {
            f0 = star_value(coef, hh + sh0, ww + sw0);
            f1 = galaxy_value(tc, NC, hh - si.m1, ww - si.m2, etab);
        } else {
            // the two densities in single precision (differences of coordinates formed in double), the moments in double
            f0 = (double)star_value_f(coef, (float)(hh + sh0), (float)(ww + sw0));
            f1 = (double)galaxy_value_f(tcf, NC, (float)(hh - si.m1), (float)(ww - si.m2));
        }
        const double En = si.c0 * f0 + si.c1 * f1;                      // E_G_s.v  (elbo_objective.jl:62-65)
        const double E2n = si.q0 * (f0 * f0) + si.q1 * (f1 * f1);
        double2 *const o = out + ((h0 - P.off_h) + (int64_t)P.H2 * (w0 - P.off_w));
        const double var = E2n - En * En;                               // var_G_s.v (:204)
        if constexpr (COH) { stc<true>(&o->x, En); stc<true>(&o->y, var); }
        else *o = make_double2(En, var);
    }
}

// value_pixels in single precision with TWO pixels per lane (idx and idx + 64 in the halves of float2 values, as in
// pixel_iter_px2): the component records' pair-interleaved fields are broadcast to both pixels (op_sel), the offsets d = x - xi
// and -log2(e)/2 d are formed once per run of prototypes, the spline runs packed on the float copy of the coefficients; the
// moments E, var are formed in double per pixel as before.
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void value_pixels_f2(int lane, const DevPatch &P, const SrcImg &si, int NC, const float *__restrict__ coef,
                                                int h_lo, int w_lo, int RH, int p0, int p1, double2 *__restrict__ out,
                                                const float *__restrict__ tcf) {
    const double sh0 = 26.0 - si.m1, sw0 = 26.0 - si.m2;
    const float rRH = 1.0f / (float)RH;
    const f2v *tp = reinterpret_cast<const f2v *>(tcf);     // per pair of components: p11, p12, p22, xi1, xi2, w0
    const int n_dev = 8 * (NC / 14);
    for (int base = p0; base < p1; base += 128) {
        int h0[2], w0[2];
        bool ok[2];
        float dxs[2], dys[2], xh[2], xw[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int i = base + 64 * q + lane;
            ok[q] = i < p1;
            int rw, rh;
            divmod_small(min(i, p1 - 1), RH, rRH, rw, rh);
            h0[q] = h_lo + rh; w0[q] = w_lo + rw;              // 0-based image coordinates
            const double hh = (double)(h0[q] + 1), ww = (double)(w0[q] + 1);
            dxs[q] = (float)(hh - si.m1); dys[q] = (float)(ww - si.m2);
            xh[q] = (float)(hh + sh0); xw[q] = (float)(ww + sw0);
        }
        // galaxy density
        const f2v dx = {dxs[0], dxs[1]}, dy = {dys[0], dys[1]};
        f2v f1 = (f2v)(0.0f);
        for (int c0 = 0; c0 < NC; c0 += (c0 < n_dev ? 8 : 6)) {
            const int len = c0 < n_dev ? 8 : 6;
            const f2v *k0 = tp + 3 * c0;
            const f2v d1 = dx - k0[3].xx, d2 = dy - k0[4].xx;
            // (as galaxy_value: the exponent from the three products of the offset, here with log2(e) folded in)
            const f2v s11 = -0.72134752044448170368f * (d1 * d1), s12 = -1.44269504088896340736f * (d1 * d2),
                      s22 = -0.72134752044448170368f * (d2 * d2);
            for (int c = c0; c < c0 + len; c += 2) {
                const f2v *k = tp + 3 * c;
                {
                    const f2v p11 = k[0].xx, p12 = k[1].xx, p22 = k[2].xx, w = k[5].xx;
                    const f2v q2 = p11 * s11 + (p12 * s12 + p22 * s22);
                    f1 += w * (f2v){__builtin_amdgcn_exp2f(q2.x), __builtin_amdgcn_exp2f(q2.y)};
                }
                {
                    const f2v p11 = k[0].yy, p12 = k[1].yy, p22 = k[2].yy, w = k[5].yy;
                    const f2v q2 = p11 * s11 + (p12 * s12 + p22 * s22);
                    f1 += w * (f2v){__builtin_amdgcn_exp2f(q2.x), __builtin_amdgcn_exp2f(q2.y)};
                }
            }
        }
        // star density: natural bicubic spline at (xh, xw), softpluslikeinv (fsm_util.jl:221-236)
        const float *cc[2];
        float fxs[2], fys[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            int ix = (int)floorf(xh[q]); ix = ix < 1 ? 1 : (ix > 50 ? 50 : ix);
            int iy = (int)floorf(xw[q]); iy = iy < 1 ? 1 : (iy > 50 ? 50 : iy);
            cc[q] = coef + (ix - 1) + CEL_COEF * (iy - 1);
            fxs[q] = xh[q] - (float)ix; fys[q] = xw[q] - (float)iy;
        }
        f2v wx[4], wy[4];
        bspline_w((f2v){fxs[0], fxs[1]}, wx); bspline_w((f2v){fys[0], fys[1]}, wy);
        f2v y = (f2v)(0.0f);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const float *ca = cc[0] + CEL_COEF * b, *cb = cc[1] + CEL_COEF * b;
            const f2v r = (f2v){ca[0], cb[0]} * wx[0] + (f2v){ca[1], cb[1]} * wx[1] + (f2v){ca[2], cb[2]} * wx[2] + (f2v){ca[3], cb[3]} * wx[3];
            y += r * wy[b];
        }
        const float f0s[2] = {y.x < 0 ? 1e-3f * __expf(y.x) : 1e-3f * (y.x + 1.0f), y.y < 0 ? 1e-3f * __expf(y.y) : 1e-3f * (y.y + 1.0f)};
        const float f1s[2] = {f1.x, f1.y};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (!ok[q]) continue;
            const double f0 = (double)f0s[q], g1 = (double)f1s[q];
            const double En = si.c0 * f0 + si.c1 * g1;                      // E_G_s.v  (elbo_objective.jl:62-65)
            const double E2n = si.q0 * (f0 * f0) + si.q1 * (g1 * g1);
            out[(h0[q] - P.off_h) + (int64_t)P.H2 * (w0[q] - P.off_w)] = make_double2(En, E2n - En * En);   // var_G_s.v (:204)
        }
    }
}

// One wavefront per work item (neighbour link t -> s2, image, chunk of the overlap): renders s2's value-only
// light on the rectangle where s2's patch (minus its last column, elbo_objective.jl:349) overlaps target t's
// patch, into s2's own patch buffer.  The items with a non-empty overlap are listed once per context
// (geometry only; celeste_ctx_create), first chunks first.  Items whose source t is not a target of this batch
// exit immediately.  Two targets that overlap the same part of s2 write identical values to the same addresses
// (benign).
#ifdef VALUE_TIMING   // debug builds (tools/variants): shader clocks per section of the value kernel
__device__ unsigned long long g_value_clk[8];
#endif
// WAVES = 4 (small, latency-bound batches; 2 in single precision, two pixels per lane): the item's up to four 64-pixel
// iterations on four wavefronts of one workgroup instead of one after the other -- same values (every pixel is computed on
// its own, by the same instructions whatever WAVES is), a quarter of the item's latency.
template <typename R, int WAVES = 1>
__global__ void __launch_bounds__(64 * WAVES)
value_kernel(const DevPatch *__restrict__ patches, const double *__restrict__ coefs,
             const SrcImg *__restrict__ srcimg, const Comp *__restrict__ comps,
             const int32_t *__restrict__ is_target, int32_t stamp, const int64_t *__restrict__ val_off,
             const int4 *__restrict__ items, int NC, int chunk_px, double2 *__restrict__ val,
             const float *__restrict__ coefs_f) {
    __shared__ double etab[64];
#ifdef VALUE_TIMING
    long long vts[4]; vts[0] = clock64();     // stamps only; the (contended) atomics all come at the very end
#define VT(k) do { vts[(k) + 1] = clock64(); } while (0)
#else
#define VT(k) do { } while (0)
#endif
    // an item = {table index of the neighbour's (source, image) entry, of the target's, chunk, target}: one 16-byte
    // load instead of a chain link -> source, link -> neighbour, item -> image | chunk
    const int4 it = items[blockIdx.x];
    const int sn = it.x, ch = it.z, t = it.w;
    if (is_target[t] != stamp) return;
    const DevPatch &P = patches[sn];
    const DevPatch &T = patches[it.y];
    // overlap rectangle in 0-based image coordinates [h_lo, h_hi) x [w_lo, w_hi)
    const int h_lo = max(P.off_h, T.off_h), h_hi = min(P.off_h + P.H2, T.off_h + T.H2);
    const int w_lo = max(P.off_w, T.off_w), w_hi = min(P.off_w + P.W2 - 1, T.off_w + T.W2);
    const int RH = h_hi - h_lo, RW = w_hi - w_lo;
    if (RH <= 0 || RW <= 0) return;
    const int npx = RH * RW;
    const int p0 = ch * chunk_px;
    if (p0 >= npx) return;
    VT(0);
    exp_table_init(etab);
    const int p1 = min(npx, p0 + chunk_px);
    const SrcImg si = srcimg[sn];
    __shared__ Comp tc[14 * CEL_MAXK];
    {
        const double *src = reinterpret_cast<const double *>(comps + (size_t)sn * NC);
        double *dst = reinterpret_cast<double *>(tc);
        for (int i = threadIdx.x; i < NC * 8; i += 64 * WAVES) dst[i] = src[i];
        __syncthreads();
    }
    __shared__ float tcf[sizeof(R) == 4 ? 6 * 14 * CEL_MAXK : 2];
    if constexpr (sizeof(R) == 4) {
        if ((int)threadIdx.x < NC) {      // pair-interleaved single-precision records (galaxy_value_f)
            const Comp k = tc[threadIdx.x];
            float *o = tcf + 12 * (threadIdx.x >> 1) + (threadIdx.x & 1);
            o[0] = (float)k.p11; o[2] = (float)k.p12; o[4] = (float)k.p22; o[6] = (float)k.xi1; o[8] = (float)k.xi2; o[10] = (float)k.w0;
        }
        __syncthreads();
    }
    VT(1);
    if constexpr (sizeof(R) == 4) {
        // single precision: ONE per-pixel arithmetic (value_pixels_f2: packed, exp2 with the pre-scaled exponent) whatever the
        // batch size -- a target's result must not depend on what else is in its launch.  WAVES wavefronts take 128 pixels each.
        const int q0 = p0 + 128 * (int)(threadIdx.x >> 6);
        if (WAVES == 1 || q0 < p1)
            value_pixels_f2(threadIdx.x & 63, P, si, NC, coefs_f + (size_t)(CELESTE_MUTANT == 3 ? 0 : P.stamp) * (CEL_COEF * CEL_COEF), h_lo, w_lo, RH,
                            WAVES == 1 ? p0 : q0, WAVES == 1 ? p1 : min(p1, q0 + 128), val + val_off[sn], tcf);
    } else if constexpr (WAVES == 1)
        value_pixels<false, R>(threadIdx.x, P, si, tc, NC, coefs, etab, h_lo, w_lo, RH, p0, p1, val + val_off[sn], tcf);
    else {
        const int q0 = p0 + 64 * (int)(threadIdx.x >> 6);
        if (q0 < p1) value_pixels<false, R>(threadIdx.x & 63, P, si, tc, NC, coefs, etab, h_lo, w_lo, RH, q0, min(p1, q0 + 64), val + val_off[sn], tcf);
    }
    VT(2);
#ifdef VALUE_TIMING
    if (threadIdx.x == 0) {
        for (int k = 0; k < 3; ++k) atomicAdd(&g_value_clk[k], (unsigned long long)(vts[k + 1] - vts[k]));
        atomicAdd(&g_value_clk[7], 1ull); atomicAdd(&g_value_clk[6], (unsigned long long)((p1 - p0 + 63) / 64));
    }
#endif
#undef VT
}

// ---------------------------------------------------------------------------------------------
// pixel_kernel
// ---------------------------------------------------------------------------------------------
// Per-pixel quantities from which every entry of the 68-double record is formed.  S: double; float in the single-precision
// mode (CELESTE_FLAG_FP32), where everything per pixel -- the terms below, the record entries, the chunk's slots in LDS --
// is fp32 and only the chunk records leave the kernel as doubles.
template <typename S>
struct PixelTermsT {
    typedef S scalar;
    S vterm, cnt_act, cnt_inact;
    S f0, f1;
    S alpha, beta, w2, w12;
    S k0, k1;
    S f0g0, f0g1, f0h0, f0h1, f0h2;
    // With dA = c0 ds + c1 dg and dB = 2 q0 f0 ds + 2 q1 f1 dg (ds / dg = gradients of the star / galaxy density) every
    // derivative entry is a combination of ds, dg and their second derivatives with per-pixel scalar weights:
    S C0s, C0g, C1s, C1g;   // rows (c0, geo), (c1, geo):  C.s ds + C.g dg
    S Q0s, Q0g, Q1s, Q1g;   // rows (q0, geo), (q1, geo)
    S Wgg, Wsg, Wss;        // (geo, geo): Wgg dg dg' + Wsg (ds dg' + dg ds') + Wss ds ds' + k1 d2g + k0 d2s
    // galaxy component sums (see pixel_kernel)
    S S0d, S1x, S1y, S1xd, S1yd, S2a, S2b, S2c, S2an, S2bn, S2cn, S2ad, S2bd, S2cd;
    S S3a, S3b, S3c, S3d, S4a, S4b, S4c, S4d, S4e;
};
typedef PixelTermsT<double> PixelTerms;

// d f1 / d(m1 m2 dev Xi11 Xi12 Xi22)
template <int G, class TT>
__device__ __forceinline__ typename TT::scalar gal_g(const TT &T) {
    typedef typename TT::scalar S;
    if constexpr (G == 0) return T.S1x;
    else if constexpr (G == 1) return T.S1y;
    else if constexpr (G == 2) return T.S0d;
    else if constexpr (G == 3) return (S)0.5 * T.S2an;
    else if constexpr (G == 4) return T.S2bn;
    else return (S)0.5 * T.S2cn;
}
// d2 f1 (upper triangle, G <= G2): Gaussian derivatives d/dm = -d/dx, d/dXi = (nu/2) d2/dx2
template <int G, int G2, class TT>
__device__ __forceinline__ typename TT::scalar gal_h(const TT &T) {
    typedef typename TT::scalar S;
    constexpr int k = G * 6 + G2;
    if constexpr (k == 0) return T.S2a;                 // m1 m1
    else if constexpr (k == 1) return T.S2b;            // m1 m2
    else if constexpr (k == 2) return T.S1xd;           // m1 dev
    else if constexpr (k == 3) return (S)0.5 * T.S3a;      // m1 Xi11
    else if constexpr (k == 4) return T.S3b;            // m1 Xi12
    else if constexpr (k == 5) return (S)0.5 * T.S3c;      // m1 Xi22
    else if constexpr (k == 7) return T.S2c;            // m2 m2
    else if constexpr (k == 8) return T.S1yd;
    else if constexpr (k == 9) return (S)0.5 * T.S3b;
    else if constexpr (k == 10) return T.S3c;
    else if constexpr (k == 11) return (S)0.5 * T.S3d;
    else if constexpr (k == 14) return (S)0.0;             // dev dev
    else if constexpr (k == 15) return (S)0.5 * T.S2ad;
    else if constexpr (k == 16) return T.S2bd;
    else if constexpr (k == 17) return (S)0.5 * T.S2cd;
    else if constexpr (k == 21) return (S)0.25 * T.S4a;    // Xi11 Xi11
    else if constexpr (k == 22) return (S)0.5 * T.S4b;
    else if constexpr (k == 23) return (S)0.25 * T.S4c;
    else if constexpr (k == 28) return T.S4c;           // Xi12 Xi12
    else if constexpr (k == 29) return (S)0.5 * T.S4d;
    else { static_assert(k == 35, "upper triangle only"); return (S)0.25 * T.S4e; }  // Xi22 Xi22
}
template <int G, class TT>
__device__ __forceinline__ typename TT::scalar star_g(const TT &T) {
    typedef typename TT::scalar S;
    if constexpr (G == 0) return T.f0g0;
    else if constexpr (G == 1) return T.f0g1;
    else return (S)0.0;
}
constexpr int hess_row(int e) {  // packed upper-triangle index (e - ACC_H0) -> row
    int i = 0, k = e - ACC_H0;
    while (k >= ZV - i) { k -= ZV - i; ++i; }
    return i;
}
constexpr int hess_col(int e) {
    int i = 0, k = e - ACC_H0;
    while (k >= ZV - i) { k -= ZV - i; ++i; }
    return i + k;
}

// entry E of the 68-double record for this pixel
template <int E, class TT>
__device__ __forceinline__ typename TT::scalar record_entry(const TT &T) {
    typedef typename TT::scalar S;
    if constexpr (E == 0) return T.vterm;
    else if constexpr (E == ACC_CNT) return T.cnt_act;
    else if constexpr (E == ACC_CNT + 1) return T.cnt_inact;
    else if constexpr (E <= ZV) {  // gradient
        constexpr int r = E - 1;
        if constexpr (r == 0) return T.alpha * T.f0;
        else if constexpr (r == 1) return T.alpha * T.f1;
        else if constexpr (r == 2) return T.w2 * T.f0 * T.f0;
        else if constexpr (r == 3) return T.w2 * T.f1 * T.f1;
        else if constexpr (r - 4 < 2) return T.k0 * star_g<r - 4>(T) + T.k1 * gal_g<r - 4>(T);
        else return T.k1 * gal_g<r - 4>(T);
    } else {
        constexpr int i = hess_row(E), j = hess_col(E);
        if constexpr (j < 2) return T.beta * (i == 0 ? T.f0 : T.f1) * (j == 0 ? T.f0 : T.f1);      // (c, c)
        else if constexpr (i < 2 && j < 4) {                                                         // (c, q)
            const S fi = i == 0 ? T.f0 : T.f1, fj = j == 2 ? T.f0 : T.f1;
            return T.w12 * fi * (fj * fj);
        } else if constexpr (j < 4) return (S)0.0;                                                      // (q, q)
        else if constexpr (i < 4) {                                                                   // (c / q, geo)
            constexpr int g2 = j - 4;
            const S xs = i == 0 ? T.C0s : (i == 1 ? T.C1s : (i == 2 ? T.Q0s : T.Q1s));
            const S xg = i == 0 ? T.C0g : (i == 1 ? T.C1g : (i == 2 ? T.Q0g : T.Q1g));
            if constexpr (g2 < 2) return xs * star_g<g2>(T) + xg * gal_g<g2>(T);
            else return xg * gal_g<g2>(T);
        } else {                                                                                      // (geo, geo)
            constexpr int g = i - 4, g2 = j - 4;
            S a = T.Wgg * gal_g<g>(T);
            if constexpr (g < 2) a += T.Wsg * star_g<g>(T);
            S v = T.k1 * gal_h<g, g2>(T) + a * gal_g<g2>(T);
            if constexpr (g2 < 2) {   // then g < 2 as well (upper triangle)
                const S f0h = (g + g2 == 0) ? T.f0h0 : ((g + g2 == 1) ? T.f0h1 : T.f0h2);
                v += T.k0 * f0h + (T.Wsg * gal_g<g>(T) + T.Wss * star_g<g>(T)) * star_g<g2>(T);
            }
            return v;
        }
    }
}

// Accumulation of the per-pixel record into the chunk's record: every lane adds its 68 entries into LDS with
// ds_add_f64, 16 slots per entry (slot = lane & 15, so the lanes l, l + 16, l + 32, l + 48 of one instruction meet in
// one slot; the LDS unit serialises them in a fixed order -- results are bitwise reproducible).  The first version
// folded the entries across lane quads with DPP exchanges and kept 17 accumulators per lane: 374 VALU instructions per
// 64 pixels (12 v_cndmask + 6 DPP moves + 4 adds per group of four entries) and 34 VGPRs; the LDS adder does the same
// work off the VALU (68 LDS instructions, ~11 cycles each per CU, a quarter of the LDS pipe's time at 8 waves per
// CU) and needs no accumulator registers.  Identically zero entries (the (q, q) block) are never touched.
#define ACC_SLOTS 16
template <int E>
constexpr bool entry_is_zero() {
    if (E > ZV && E < ACC_CNT) { const int i = hess_row(E), j = hess_col(E); return i >= 2 && i < 4 && j < 4; }
    return false;
}
// (The slots are fp64 in the single-precision mode too -- the entry is widened for the add: ds_add_f32 turned out four
// times slower than ds_add_f64 on gfx950 in this access pattern, 35.9 against 8.2 ms for the config-5 sweep, and its
// sums differed between two launches.)
template <int MODE, int E, class TT>
__device__ __forceinline__ void accum_entries(const TT &T, double *__restrict__ slot) {
    constexpr bool hess_only = E > ZV && E < ACC_CNT;
    if constexpr (!(MODE == 1 && hess_only) && !entry_is_zero<E>())
        __hip_atomic_fetch_add(slot + ACC_SLOTS * E, (double)record_entry<E>(T), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if constexpr (E + 1 < ACC_N) accum_entries<MODE, E + 1>(T, slot);
}

// Single-precision mode: FOUR consecutive entries per LDS add.  The slot of a lane is lane & 15, so the lanes l, l + 16,
// l + 32, l + 48 of one ds_add_f64 meet in one address and the LDS unit serialises them; at the fp32 kernel's pace (an
// iteration is half as long as the fp64 kernel's) the 65 four-way-conflicting adds per iteration are what its wavefronts
// queue on.  Here the four rows of 16 lanes are folded in registers first -- v_permlane16_swap / v_permlane32_swap put the
// partial sums of entries E .. E + 3 into rows 0 .. 3 of ONE register (three swaps and three fp32 adds for four entries) --
// and row r of the wavefront adds its entry E + r: 17 conflict-free adds instead of 65 four-way ones.  The sum of a slot's
// four lanes is formed in fp32 in a fixed order, (l + l16) + (l32 + l48): reproducible, inside the mode's 1e-4 by orders
// of magnitude.  Entries a mode does not produce, and the identically zero ones, enter as 0.
template <int MODE, int E>
constexpr bool group_is_empty() {    // none of E .. E + 3 is produced
    bool any = false;
    for (int k = 0; k < 4; ++k) {
        const int e = E + k;
        if (e >= ACC_N) continue;
        const bool hess_only = e > ZV && e < ACC_CNT;
        if (MODE == 1 && hess_only) continue;
        any = true;
    }
    return !any;
}
__device__ __forceinline__ float fold_rows_pair16(float a, float b) {   // rows (a01, b01, a23, b23)
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// (the adds themselves: accum_entries_rows2, two pixels per lane -- the only user since round 5)

// the same in two steps: the values (pinned in registers by an empty asm, so that they are formed where this is called),
// then the adds
template <int MODE, int E, class TT>
__device__ __forceinline__ void form_entries(const TT &T, typename TT::scalar (&ent)[ACC_N]) {
    constexpr bool hess_only = E > ZV && E < ACC_CNT;
    if constexpr (!(MODE == 1 && hess_only) && !entry_is_zero<E>()) { ent[E] = record_entry<E>(T); asm volatile("" : "+v"(ent[E])); }
    if constexpr (E + 1 < ACC_N) form_entries<MODE, E + 1>(T, ent);
}
template <int MODE, int E, typename S>
__device__ __forceinline__ void add_entries(const S (&ent)[ACC_N], double *__restrict__ slot) {
    constexpr bool hess_only = E > ZV && E < ACC_CNT;
    if constexpr (!(MODE == 1 && hess_only) && !entry_is_zero<E>())
        __hip_atomic_fetch_add(slot + ACC_SLOTS * E, (double)ent[E], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if constexpr (E + 1 < ACC_N) add_entries<MODE, E + 1>(ent, slot);
}

// Split variant (CELESTE_FLAG_SPLIT): instead of folding, every pixel's 68-entry record goes to HBM, entry-major
// inside a 64-pixel tile (rec[e * 64 + lane]) so that each store instruction writes one contiguous 512-byte row.
template <int E, class TT>
__device__ __forceinline__ void store_entries(const TT &T, double *__restrict__ tile_lane) {
    __builtin_nontemporal_store((double)record_entry<E>(T), tile_lane + E * 64);
    if constexpr (E + 1 < ACC_N) store_entries<E + 1>(T, tile_lane);
}

// Component record in the arithmetic type of the pixel math (double, or float for CELESTE_FLAG_FP32)
template <typename R>
struct CompR { R p11, p12, p22, w0, wd, nu, xi1, xi2; };

template <typename R> __device__ __forceinline__ R exp_np(R x, const double *tab);
template <> __device__ __forceinline__ double exp_np<double>(double x, const double *tab) { return exp_nonpos(x, tab); }
template <> __device__ __forceinline__ float exp_np<float>(float x, const double *) { return __expf(x); }  // v_exp_f32
template <typename R> __device__ __forceinline__ R fma_r(R a, R b, R c);
template <> __device__ __forceinline__ double fma_r<double>(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <> __device__ __forceinline__ float fma_r<float>(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// Galaxy: 14 * psf_K bivariate normals (accum_galaxy_pos!, fsm_util.jl:255-346).  The reference's
// per-component chain get_bvn_derivs! -> transform_bvn_derivs! is replaced by its closed form: the
// derivatives of a Gaussian density with respect to its mean are Hermite polynomials in
// (u, v) = P (x - mu), and d/dSigma = (1/2) d2/dx2, so every quantity the Hessian needs is a
// weighted sum over components of spatial derivatives up to order 4 (24 sums instead of 1+6+21,
// and no 3x3 transforms inside the loop).  Weights: w0 = z theta_i, wd = +-z, and their products with
// nu, nu^2.  (dx, dy) = pixel - m_pos.  Returns sum f; fills the S* members of T.
// ---- explicit LDS reads of a component record (two register sets, ping-pong) -------------------------------------------
typedef double dbl2 __attribute__((ext_vector_type(2)));
struct LdsComp { dbl2 a, b, c, d, e, f; };   // {p11, p12} {p22, w0} {wd, nu} | {-2 p12, -3 p11} {-3 p12, -3 p22} {-4 p12, -2 p12^2}
__device__ __forceinline__ unsigned lds_addr(const void *p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const void *)p;
}
__device__ __forceinline__ void lds_issue_comp(LdsComp &r, unsigned addr, unsigned addrx) {
    asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:16\n\tds_read_b128 %2, %6 offset:32\n\t"
                 "ds_read_b128 %3, %7\n\tds_read_b128 %4, %7 offset:16\n\tds_read_b128 %5, %7 offset:32"
                 : "=&v"(r.a), "=&v"(r.b), "=&v"(r.c), "=&v"(r.d), "=&v"(r.e), "=&v"(r.f) : "v"(addr), "v"(addrx) : "memory");
}
// the same with the component's position inside its run of prototypes as immediate offsets (the loop over a run fully
// unrolled): no per-component address arithmetic, no v_mov of a scalar address into a VGPR
template <int OFF, int OFFX>
__device__ __forceinline__ void lds_issue_comp_imm(LdsComp &r, unsigned addr, unsigned addrx) {
    asm volatile("ds_read_b128 %0, %6 offset:%8\n\tds_read_b128 %1, %6 offset:%9\n\tds_read_b128 %2, %6 offset:%10\n\t"
                 "ds_read_b128 %3, %7 offset:%11\n\tds_read_b128 %4, %7 offset:%12\n\tds_read_b128 %5, %7 offset:%13"
                 : "=&v"(r.a), "=&v"(r.b), "=&v"(r.c), "=&v"(r.d), "=&v"(r.e), "=&v"(r.f) : "v"(addr), "v"(addrx),
                   "i"(OFF), "i"(OFF + 16), "i"(OFF + 32), "i"(OFFX), "i"(OFFX + 16), "i"(OFFX + 32) : "memory");
}
__device__ __forceinline__ void lds_wait_comp(LdsComp &r) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r.a), "+v"(r.b), "+v"(r.c), "+v"(r.d), "+v"(r.e), "+v"(r.f));
}
// The multiples of a component's precision matrix that the third- and fourth-order Hermite polynomials need, formed once
// per workgroup next to the staged record (COMPX doubles per component in LDS): with them u^2 v - p11 v - 2 p12 u =
// fma(v, ha, u * (-2 p12)) etc. are one multiply and one FMA, where -2 u, -3 ha, -3 hc, -2 hb each cost an instruction of
// their own -- 5 of the loop's 72 VALU instructions per component.
// (round 4: two more for the xxyy polynomial in the form ha hc - 4 p12 hb - 2 p12^2 -- two FMAs instead of a multiply and two)
#define COMPX 6
__device__ __forceinline__ void comp_extra(const Comp &k, double *__restrict__ x) {
    x[0] = -2.0 * k.p12; x[1] = -3.0 * k.p11; x[2] = -3.0 * k.p12; x[3] = -3.0 * k.p22;
    x[4] = -4.0 * k.p12; x[5] = -2.0 * (k.p12 * k.p12);
}
// The quadratic form is hd1 u + hd2 v with hd = -d / 2 formed once per run of prototypes (one multiply per
// component less, same bits: a factor 1/2 commutes with rounding); the runs of 8 / 6 prototypes are fully unrolled,
// records read at immediate offsets from the run's base (two v_mov and the scalar address arithmetic per component less).
// Together 138 -> 132 VALU per two components, 0.543 -> 0.531 ms on the bench field.

template <int MODE, typename R>
__device__ __forceinline__ double galaxy_sums(const CompR<R> *tc, int n_dev, int nc, R dx, R dy, R dev, const double *etab,
                                              PixelTerms &T, const double *tcx = nullptr) {
    if constexpr (MODE == 2) {
        // The six sums that exist in an f-weighted (w0 = z theta_i) and a d-weighted (wd = +-z) version -- order 0, order
        // 1 (x, y) and the nu-weighted order 2 (xx, xy, yy) -- are accumulated ONCE, d-weighted, per profile type: the
        // components are stored type 0 (de Vaucouleurs, n_dev of them) first, so two loops over the same body fill U0
        // and U1, and afterwards  f-sum = theta_0 U0 - theta_1 U1,  d-sum = U0 + U1  (U1 carries the minus sign of wd).
        // 18 accumulations per component instead of 24.
        R U0[6] = {0, 0, 0, 0, 0, 0}, U1[6] = {0, 0, 0, 0, 0, 0};
        R S2a = 0, S2b = 0, S2c = 0, S3a = 0, S3b = 0, S3c = 0, S3d = 0, S4a = 0, S4b = 0, S4c = 0, S4d = 0, S4e = 0;
        R hd1 = 0, hd2 = 0;   // -d / 2 of the current run of prototypes
        (void)hd1; (void)hd2;
        // GW (a std::integral_constant<bool>): the sums of orders 2 .. 4 are accumulated with the d-weights g = wd e as well
        // (the de Vaucouleurs loop: f = theta_0 g there, applied ONCE to the twelve sums when the loop ends -- three weight
        // products per component instead of five)
        auto body_regs = [&](R p11, R p12, R p22, R w0, R wd, R nu, R m2p12, R m3p11, R m3p12, R m3p22, R m4p12, R m2p12sq,
                             R (&U)[6], R d1, R d2, auto GW, auto &&after_exp_issue) {
            constexpr bool gw = decltype(GW)::value;
            struct { R p11, p12, p22, w0, wd, nu; } k = {p11, p12, p22, w0, wd, nu};
            const R u = k.p11 * d1 + k.p12 * d2, v = k.p12 * d1 + k.p22 * d2;
            // exp_nonpos in two halves: the table entry is requested as soon as its index is known, the Hermite
            // polynomials (which do not depend on the exponential) are evaluated while it is on its way
            R xr, tj = 0, pe = 0;
            int ni = 0;
            if constexpr (sizeof(R) == 8) {
                double x = __builtin_fma(hd1, u, hd2 * v);        // hd = -d / 2, formed once per run of prototypes
                // (v_rndne_f64 / v_cvt_i32_f64 / v_ldexp_f64 issue FASTER than v_fma_f64 on this chip -- 4.6 / 4.7 / 5.1 against
                // 5.7 cycles per wave instruction, tools/fp64_rate_probe.hip -- so the magic-number rounding and exponent
                // arithmetic that replace them with fp64 adds and 32-bit integer operations lost 2 %: measured, removed)
                const double n = rint(x * EXP_INV_STEP);
                ni = (int)n;
                // one FMA with ln2 / T rounded to double instead of the high / low pair: the product n x step is exact inside
                // the FMA, what is left is n times the constant's representation error -- 3.6e-19 n, i.e. 2e-15 relative on
                // exp(x) at x = -50 (n = 4600) and 2e-14 at -700, against the polynomial's 3.9e-14
                xr = __builtin_fma(n, -EXP_STEP_FULL, x);
                tj = etab[ni & (EXP_T - 1)];
            } else xr = (R)-0.5 * (d1 * u + d2 * v);
            after_exp_issue();
            const R ha = fma_r<R>(u, u, -k.p11), hb = fma_r<R>(u, v, -k.p12), hc = fma_r<R>(v, v, -k.p22);
            // third order: u^3 - 3 p11 u, u^2 v - p11 v - 2 p12 u, u v^2 - p22 u - 2 p12 v, v^3 - 3 p22 v
            const R h3a = u * fma_r<R>((R)-2.0, k.p11, ha);
            const R h3b = fma_r<R>(v, ha, u * m2p12);
            const R h3c = fma_r<R>(u, hc, v * m2p12);
            const R h3d = v * fma_r<R>((R)-2.0, k.p22, hc);
            // fourth order
            const R h4a = fma_r<R>(u, h3a, ha * m3p11);
            const R h4b = fma_r<R>(v, h3a, ha * m3p12);
            const R h4c = fma_r<R>(ha, hc, fma_r<R>(hb, m4p12, m2p12sq));   // u^2 v^2 - p22 u^2 - p11 v^2 - 4 p12 u v + p11 p22 + 2 p12^2
            const R h4d = fma_r<R>(u, h3d, hc * m3p12);
            const R h4e = fma_r<R>(v, h3d, hc * m3p22);
            R e;
            if constexpr (sizeof(R) == 8) {
                pe = exp_poly(xr);
                __builtin_amdgcn_sched_barrier(0);   // nothing that needs the table entry moves above this point
                e = ldexp(pe * tj, ni >> EXP_TAB_LOG2);   // eval_bvn_pdf!
            } else e = exp_np<R>(xr, etab);
            const R g = k.wd * e, gn = g * k.nu;
            R f, fn, fnn;
            if constexpr (gw) { f = g; fn = gn; fnn = gn * k.nu; }
            else { f = k.w0 * e; fn = f * k.nu; fnn = fn * k.nu; }
            U[0] += g; U[1] = fma_r<R>(u, g, U[1]); U[2] = fma_r<R>(v, g, U[2]);
            U[3] = fma_r<R>(ha, gn, U[3]); U[4] = fma_r<R>(hb, gn, U[4]); U[5] = fma_r<R>(hc, gn, U[5]);
            S2a = fma_r<R>(ha, f, S2a); S2b = fma_r<R>(hb, f, S2b); S2c = fma_r<R>(hc, f, S2c);
            S3a = fma_r<R>(h3a, fn, S3a); S3b = fma_r<R>(h3b, fn, S3b);
            S3c = fma_r<R>(h3c, fn, S3c); S3d = fma_r<R>(h3d, fn, S3d);
            S4a = fma_r<R>(h4a, fnn, S4a); S4b = fma_r<R>(h4b, fnn, S4b);
            S4c = fma_r<R>(h4c, fnn, S4c); S4d = fma_r<R>(h4d, fnn, S4d);
            S4e = fma_r<R>(h4e, fnn, S4e);
        };
        auto body = [&](int c, R (&U)[6], R d1, R d2) {
            const CompR<R> k = tc[c];
            body_regs(k.p11, k.p12, k.p22, k.w0, k.wd, k.nu, (R)tcx[COMPX * c], (R)tcx[COMPX * c + 1], (R)tcx[COMPX * c + 2],
                      (R)tcx[COMPX * c + 3], (R)tcx[COMPX * c + 4], (R)tcx[COMPX * c + 5], U, d1, d2, std::false_type(), []() {});
        };
        // the de Vaucouleurs sums of orders 2 .. 4 were accumulated d-weighted (GW): f = theta_0 g
        auto scale_dev = [&](R th) {
            S2a *= th; S2b *= th; S2c *= th; S3a *= th; S3b *= th; S3c *= th; S3d *= th;
            S4a *= th; S4b *= th; S4c *= th; S4d *= th; S4e *= th;
        };
        (void)scale_dev;
        // runs of 8 (de Vaucouleurs) / 6 (exponential) prototypes share a PSF component, i.e. the offset xiBar_k
        if constexpr (sizeof(R) == 8) {
            // The record of component c + 1 is requested while component c is computed (two register sets, the loop
            // unrolled by two): the compiler neither rotates the loop nor leaves a hand-hoisted load where it is put,
            // so the reads are volatile asm, and every set passes through the "+v" operands of an explicit s_waitcnt
            // before it is used (free when the data has already arrived under the exponential's table read).
            const unsigned base = lds_addr(tc), basex = lds_addr(tcx);
            LdsComp ra, rb;
            lds_issue_comp(ra, base, basex);
            // runs of 8 / 6 prototypes fully unrolled: component j of a run is read at (run base) + 64 j, an immediate; the
            // request that follows a run's last component lands on the next run's first record (or, after the very last
            // run, on the 96 bytes behind the tables -- inside the workgroup's LDS, never used)
            auto half_imm = [&](LdsComp &k, LdsComp &nxt, auto JN, unsigned vb, unsigned vbx, R (&U)[6], R d1, R d2, auto GW) {
                constexpr int jn = decltype(JN)::value;
                lds_wait_comp(k);
                body_regs(k.a.x, k.a.y, k.b.x, k.b.y, k.c.x, k.c.y, k.d.x, k.d.y, k.e.x, k.e.y, k.f.x, k.f.y, U, d1, d2, GW, [&]() {
                    lds_issue_comp_imm<64 * jn, COMPX * 8 * jn>(nxt, vb, vbx);
                });
            };
#define IC(n) std::integral_constant<int, n>()
            // (do-while, here and below: psf_K >= 1, so each profile type has at least one run; with a `for` the compiler keeps a
            // zeroed copy of every sum for the path around the loop -- PX_DOWHILE 0 restores it for the A/B)
#ifndef PX_DOWHILE
#define PX_DOWHILE 1
#endif
#if PX_DOWHILE
#define RUN_LOOP(init, cond, step) { init; do {
#define RUN_LOOP_END(cond, step) step; } while (cond); }
#else
#define RUN_LOOP(init, cond, step) for (init; cond; step) {
#define RUN_LOOP_END(cond, step) }
#endif
            RUN_LOOP(int c0 = 0, c0 < n_dev, c0 += 8)
                const R d1 = dx - tc[c0].xi1, d2 = dy - tc[c0].xi2;
                hd1 = (R)-0.5 * d1; hd2 = (R)-0.5 * d2;
                const unsigned vb = base + 64u * (unsigned)c0, vbx = basex + (unsigned)(COMPX * 8) * (unsigned)c0;
                const std::true_type GWT;
                half_imm(ra, rb, IC(1), vb, vbx, U0, d1, d2, GWT); half_imm(rb, ra, IC(2), vb, vbx, U0, d1, d2, GWT);
                half_imm(ra, rb, IC(3), vb, vbx, U0, d1, d2, GWT); half_imm(rb, ra, IC(4), vb, vbx, U0, d1, d2, GWT);
                half_imm(ra, rb, IC(5), vb, vbx, U0, d1, d2, GWT); half_imm(rb, ra, IC(6), vb, vbx, U0, d1, d2, GWT);
                half_imm(ra, rb, IC(7), vb, vbx, U0, d1, d2, GWT); half_imm(rb, ra, IC(8), vb, vbx, U0, d1, d2, GWT);
            RUN_LOOP_END(c0 < n_dev, c0 += 8)
            scale_dev(dev);
            RUN_LOOP(int c0 = n_dev, c0 < nc, c0 += 6)
                const R d1 = dx - tc[c0].xi1, d2 = dy - tc[c0].xi2;
                hd1 = (R)-0.5 * d1; hd2 = (R)-0.5 * d2;
                const unsigned vb = base + 64u * (unsigned)c0, vbx = basex + (unsigned)(COMPX * 8) * (unsigned)c0;
                const std::false_type GWF;
                half_imm(ra, rb, IC(1), vb, vbx, U1, d1, d2, GWF); half_imm(rb, ra, IC(2), vb, vbx, U1, d1, d2, GWF);
                half_imm(ra, rb, IC(3), vb, vbx, U1, d1, d2, GWF); half_imm(rb, ra, IC(4), vb, vbx, U1, d1, d2, GWF);
                half_imm(ra, rb, IC(5), vb, vbx, U1, d1, d2, GWF); half_imm(rb, ra, IC(6), vb, vbx, U1, d1, d2, GWF);
            RUN_LOOP_END(c0 < nc, c0 += 6)
#undef RUN_LOOP
#undef RUN_LOOP_END
#undef IC
            lds_wait_comp(ra);   // the last (unused) request must land before its registers are reused
        } else
        {
        for (int c0 = 0; c0 < n_dev; c0 += 8) {
            const R d1 = dx - tc[c0].xi1, d2 = dy - tc[c0].xi2;
            hd1 = (R)-0.5 * d1; hd2 = (R)-0.5 * d2;
            for (int c = c0; c < c0 + 8; ++c) body(c, U0, d1, d2);
        }
        for (int c0 = n_dev; c0 < nc; c0 += 6) {
            const R d1 = dx - tc[c0].xi1, d2 = dy - tc[c0].xi2;
            hd1 = (R)-0.5 * d1; hd2 = (R)-0.5 * d2;
            for (int c = c0; c < c0 + 6; ++c) body(c, U1, d1, d2);
        }
        }
        const R th0 = dev, th1 = (R)1.0 - dev;
        T.S0d = U0[0] + U1[0]; T.S1xd = U0[1] + U1[1]; T.S1yd = U0[2] + U1[2];
        T.S2ad = U0[3] + U1[3]; T.S2bd = U0[4] + U1[4]; T.S2cd = U0[5] + U1[5];
        T.S1x = th0 * U0[1] - th1 * U1[1]; T.S1y = th0 * U0[2] - th1 * U1[2];
        T.S2an = th0 * U0[3] - th1 * U1[3]; T.S2bn = th0 * U0[4] - th1 * U1[4]; T.S2cn = th0 * U0[5] - th1 * U1[5];
        T.S2a = S2a; T.S2b = S2b; T.S2c = S2c;
        T.S3a = S3a; T.S3b = S3b; T.S3c = S3c; T.S3d = S3d;
        T.S4a = S4a; T.S4b = S4b; T.S4c = S4c; T.S4d = S4d; T.S4e = S4e;
        return (double)(th0 * U0[0] - th1 * U1[0]);
    } else {
        R S0 = 0, S0d = 0, S1x = 0, S1y = 0, S2an = 0, S2bn = 0, S2cn = 0;
        for (int c = 0; c < nc; ++c) {
            const CompR<R> k = tc[c];
            const R d1 = dx - k.xi1, d2 = dy - k.xi2;
            const R u = k.p11 * d1 + k.p12 * d2, v = k.p12 * d1 + k.p22 * d2;
            const R e = exp_np<R>((R)-0.5 * (d1 * u + d2 * v), etab);   // eval_bvn_pdf!
            const R f = k.w0 * e, fd = k.wd * e, fn = f * k.nu;
            const R ha = fma_r<R>(u, u, -k.p11), hb = fma_r<R>(u, v, -k.p12), hc = fma_r<R>(v, v, -k.p22);
            S0 += f; S0d += fd;
            S1x = fma_r<R>(u, f, S1x); S1y = fma_r<R>(v, f, S1y);
            S2an = fma_r<R>(ha, fn, S2an); S2bn = fma_r<R>(hb, fn, S2bn); S2cn = fma_r<R>(hc, fn, S2cn);
        }
        T.S0d = S0d; T.S1x = S1x; T.S1y = S1y; T.S2an = S2an; T.S2bn = S2bn; T.S2cn = S2cn;
        return (double)S0;
    }
}

// Single-precision variant of galaxy_sums (CELESTE_FLAG_FP32) with two components per instruction: the records of
// components c and c + 1 sit in the two halves of float2 values, so the whole Hermite chain runs on v_pk_fma_f32 /
// v_pk_mul_f32 (NC = 14 psf_K is even); only the exponential is evaluated per half.  The two halves of every sum are
// added at the end.  Same arithmetic per component as galaxy_sums<MODE, float>.
#define PKSLOTS 12   // 8-byte slots per PAIR of components in the fp32 record table: the eight fields of Comp, then -2 p12, -3 p11, -3 p12, -3 p22
template <int MODE, class TT>
__device__ __forceinline__ typename TT::scalar galaxy_sums_pk(const CompR<float> *tc, int n_dev, int nc, float dx, float dy,
                                                              float dev, TT &T) {
    typedef typename TT::scalar S;
    const f2v z = (f2v)(0.0f);
    const f2v dxx = (f2v)(dx), dyy = (f2v)(dy);
    // the records are staged pair-interleaved (pixel_kernel's prologue): field i of components c, c + 1 sits in the two
    // halves of one 8-byte slot, so the packed operands come out of the LDS reads as they are (13 v_mov per trip saved)
    const f2v *tp = reinterpret_cast<const f2v *>(tc);
    // (Hessian sums only: the split variant, pixel_kernel<3, float>, is the one user left -- the fused single-precision modes
    // run galaxy_sums_px2, two pixels per lane)
    static_assert(MODE == 2, "split variant only");
    {
        // as in galaxy_sums: the six sums that exist f-weighted and d-weighted are accumulated once, d-weighted, per profile
        // type (U0: de Vaucouleurs, U1: exponential, carrying wd's minus sign); pairs never straddle the types (8 psf_K and
        // 6 psf_K are even).  18 packed accumulations per pair instead of 24, and the Hermite polynomials take their
        // multiples of the precision matrix from the record (comp_extra's constants).
        f2v U0[6] = {z, z, z, z, z, z}, U1[6] = {z, z, z, z, z, z};
        f2v S2a = z, S2b = z, S2c = z, S3a = z, S3b = z, S3c = z, S3d = z, S4a = z, S4b = z, S4c = z, S4d = z, S4e = z;
        auto pair = [&](int c, f2v (&U)[6]) {
            const f2v *k = tp + (PKSLOTS / 2) * c;
            const f2v p11 = k[0], p12 = k[1], p22 = k[2], w0 = k[3], wd = k[4], nu = k[5], xi1 = k[6], xi2 = k[7];
            const f2v m2p12 = k[8], m3p11 = k[9], m3p12 = k[10], m3p22 = k[11];
            const f2v d1 = dxx - xi1, d2 = dyy - xi2;
            const f2v u = p11 * d1 + p12 * d2, v = p12 * d1 + p22 * d2;
            const f2v q = -0.5f * (d1 * u + d2 * v);
            const f2v e = {__expf(q.x), __expf(q.y)};
            const f2v f = w0 * e, g = wd * e, fn = f * nu, gn = g * nu, fnn = fn * nu;
            const f2v ha = u * u - p11, hb = u * v - p12, hc = v * v - p22;
            U[0] += g; U[1] += u * g; U[2] += v * g;
            U[3] += ha * gn; U[4] += hb * gn; U[5] += hc * gn;
            S2a += ha * f; S2b += hb * f; S2c += hc * f;
            const f2v h3a = u * (ha - 2.0f * p11), h3b = v * ha + u * m2p12, h3c = u * hc + v * m2p12, h3d = v * (hc - 2.0f * p22);
            S3a += h3a * fn; S3b += h3b * fn; S3c += h3c * fn; S3d += h3d * fn;
            const f2v h4a = u * h3a + ha * m3p11, h4b = v * h3a + ha * m3p12, h4c = u * h3c + (hb * m2p12 - hc * p11),
                      h4d = u * h3d + hc * m3p12, h4e = v * h3d + hc * m3p22;
            S4a += h4a * fnn; S4b += h4b * fnn; S4c += h4c * fnn; S4d += h4d * fnn; S4e += h4e * fnn;
        };
        for (int c = 0; c < n_dev; c += 2) pair(c, U0);
        for (int c = n_dev; c < nc; c += 2) pair(c, U1);
        const S th0 = (S)dev, th1 = (S)1.0 - (S)dev;
#define PKH(v) ((S)(v).x + (S)(v).y)
        const S u0[6] = {PKH(U0[0]), PKH(U0[1]), PKH(U0[2]), PKH(U0[3]), PKH(U0[4]), PKH(U0[5])};
        const S u1[6] = {PKH(U1[0]), PKH(U1[1]), PKH(U1[2]), PKH(U1[3]), PKH(U1[4]), PKH(U1[5])};
        T.S0d = u0[0] + u1[0]; T.S1xd = u0[1] + u1[1]; T.S1yd = u0[2] + u1[2];
        T.S2ad = u0[3] + u1[3]; T.S2bd = u0[4] + u1[4]; T.S2cd = u0[5] + u1[5];
        T.S1x = th0 * u0[1] - th1 * u1[1]; T.S1y = th0 * u0[2] - th1 * u1[2];
        T.S2an = th0 * u0[3] - th1 * u1[3]; T.S2bn = th0 * u0[4] - th1 * u1[4]; T.S2cn = th0 * u0[5] - th1 * u1[5];
        T.S2a = PKH(S2a); T.S2b = PKH(S2b); T.S2c = PKH(S2c);
        T.S3a = PKH(S3a); T.S3b = PKH(S3b); T.S3c = PKH(S3c); T.S3d = PKH(S3d);
        T.S4a = PKH(S4a); T.S4b = PKH(S4b); T.S4c = PKH(S4c); T.S4d = PKH(S4d); T.S4e = PKH(S4e);
        return th0 * u0[0] - th1 * u1[0];
    }
#undef PKH
}


// Per-pixel inputs of the pixel term: the pixel itself, sky + the pre-rendered light of the covering neighbours, the
// per-row calibration; `valid` = the pixel is visited (elbo_objective.jl:445, 459).
struct PixelInputs {
    double x, Ebar, Vbar, lgx, iota, log_iota;   // lgx: several active sources only (else the lift subtracts the patch's sum)
    int n_inact;
    bool valid, dup;
};
// COHV: the neighbours' light was rendered by other workgroups of the SAME launch (joint dataflow, fused_kernels.h): its
// loads go past the non-coherent caches
template <bool MULTI, bool COHV = false>
__device__ __forceinline__ PixelInputs load_pixel_inputs(
        const DevImage &img, const DevPatch &P, const DevPatch *__restrict__ patches, const uint8_t *__restrict__ bitmaps,
        const int32_t *__restrict__ nbr_idx, const int32_t *__restrict__ nv, int64_t nb0, int64_t nb1,
        const int64_t *__restrict__ val_off,
        const double2 *__restrict__ val, const int32_t *__restrict__ active_rank, int my_rank, int N, int n, int H2,
        int h, int w, int h2, int w2, bool in_range) {
    PixelInputs I;
    const size_t gi = (size_t)(h - 1) + (size_t)img.H * (w - 1);
    // coalesced (along h) loads of every per-pixel input, issued together
    const float xf = img.pixels[gi];
#if CELESTE_MUTANT == 2
    const float skyf = img.sky[(size_t)(w - 1) + (size_t)img.W * (h - 1)];
#else
    const float skyf = img.sky[gi];
#endif
#if CELESTE_MUTANT == 1
    const int irow = min(w, img.H);
#else
    const int irow = h;   // iota is per ROW: img.nelec_per_nmgy[h] (elbo_objective.jl:374-385)
#endif
    I.iota = (double)img.iota[irow - 1];
    I.log_iota = img.log_iota[irow - 1];
    bool valid = in_range && !isnan(xf);                // elbo_objective.jl:459
    if (P.bitmap_off >= 0) valid = valid && bitmaps[P.bitmap_off + h2 + (int64_t)H2 * w2] != 0;  // :445
    double Ebar = (double)skyf;  // epsilon + neighbours
    double Vbar = 0.0;
    int n_inact = 0;
    bool dup = false;

    // ---- neighbours: gather their pre-rendered (E_G_s.v, var_G_s.v) ----
    for (int64_t q = nb0; q < nb1; ++q) {
        const int s2 = nbr_idx[q];
        const int v2 = nv ? nv[q - nb0] : s2 * N + n;    // the neighbour's table entry for this image (visit lists: -1 = none)
        if (v2 < 0) continue;
        const DevPatch &Q = patches[v2];
        const int ph2 = h - Q.off_h, pw2 = w - Q.off_w;  // 1-based in the neighbour's patch
        bool in = valid & (ph2 >= 1) & (ph2 <= Q.H2) & (pw2 >= 1) & (pw2 < Q.W2);  // strict: elbo_objective.jl:349
        if (in && Q.bitmap_off >= 0) in = bitmaps[Q.bitmap_off + (ph2 - 1) + (int64_t)Q.H2 * (pw2 - 1)] != 0;
        int r2 = -1;
        if (MULTI) {
            r2 = active_rank[s2];
            if (r2 >= 0 && r2 < my_rank) {   // does the earlier active source visit this pixel (last column included)?
                bool vis = valid & (ph2 >= 1) & (ph2 <= Q.H2) & (pw2 >= 1) & (pw2 <= Q.W2);
                if (vis && Q.bitmap_off >= 0) vis = bitmaps[Q.bitmap_off + (ph2 - 1) + (int64_t)Q.H2 * (pw2 - 1)] != 0;
                dup |= vis;
            }
        }
        if (in) {
            const double2 *const pv = val + (val_off[v2] + (ph2 - 1) + (int64_t)Q.H2 * (pw2 - 1));
            const double2 ev = COHV ? make_double2(ldc<true>(&pv->x), ldc<true>(&pv->y)) : *pv;
            Ebar += ev.x;
            Vbar += ev.y;
            n_inact += MULTI ? (r2 < 0) : 1;
        }
    }
    if (MULTI && dup) n_inact = 0;
    I.x = (double)xf; I.Ebar = Ebar; I.Vbar = Vbar; I.n_inact = n_inact; I.valid = valid; I.dup = dup;
    // several active sources: a pixel two of them cover is counted by the earlier one only, so the term stays per pixel
    I.lgx = (MULTI && valid && !dup) ? lgamma((double)xf + 1.0) : 0.0;
    return I;
}

// MODE 0: value only; MODE 1: value + gradient sums; MODE 2: value + gradient + Hessian sums;
// MODE 3: as MODE 2 but the per-pixel records are written to HBM for record_sum_kernel (split variant)
#ifndef PIXEL_WAVES
#define PIXEL_WAVES 2  // waves per SIMD the register allocator must allow (256 VGPRs, no scratch; 3 waves with a dozen
                       // loop invariants in scratch measured 3 % slower)
#endif

// What one 64-pixel iteration of the pixel loop needs besides its position: the (target, image) patch, the target's
// per-image tables (in LDS) and the neighbour look-up.  Filled once per work item by pixel_kernel, once per queue item by
// the fused optimiser kernel (optim_fused_kernel): both run the SAME iteration body, pixel_iter.
template <typename R>
struct PixWork {
    const DevImage *img; const DevPatch *P; const DevPatch *patches; const uint8_t *bitmaps;
    const int32_t *nbr_idx; const int32_t *nv; int64_t nb0, nb1;
    const int64_t *val_off; const double2 *val; const int32_t *active_rank;
    int my_rank, N, n, NC, v;
    SrcImg si;                       // the target's pixel-space position and brightness moments for this image
    const Comp *tc;                  // its 14 psf_K components (LDS)
    const double *tcx;               // comp_extra of each (LDS; Hessian mode, fp64 loop)
    const CompR<R> *tcr;             // the same in the arithmetic type of the component loop (LDS)
    const double *etab;              // 2^(j/64) table (LDS)
    const double *tcoef;             // star spline coefficients of the patch's stamp
    const float *tcoef_f;            // the same rounded to float (single-precision mode: 4-byte loads, no conversions)
    const int64_t *tile_off; double *rec;   // split variant only
};

// One iteration of the pixel loop: lanes = the 64 pixels idx = base + lane < p1 of the patch (h fastest).  MODE 0 adds
// into a[3]; MODE 1 / 2 add the pixel's record entries into the 16 slots per entry at `slot` (accum_entries) after
// calling gate() -- a no-op in pixel_kernel, where one wave runs the iterations of a chunk in turn; the fused optimiser
// kernel gives the four iterations of a chunk to four waves and uses the gate to let them ADD in iteration order, which
// makes its chunk records bit-identical to pixel_kernel's.
template <int MODE, typename R, bool MULTI, bool GATED = false, bool COHV = false, class Gate>
__device__ __forceinline__ void pixel_iter(const PixWork<R> &W, int base, int p1, int lane, double *__restrict__ slot,
                                           double (&a)[3], Gate &&gate) {
    constexpr int GM = MODE == 3 ? 2 : MODE;  // MODE 3 = MODE 2 sums, per-pixel records stored instead of folded
    const DevImage &img = *W.img;
    const DevPatch &P = *W.P;
    const SrcImg &si = W.si;
    const int H2 = P.H2, W2 = P.W2;
    // index offsets of the star spline: itp[h - m1 + 26, w - m2 + 26]
    const double sh0 = 26.0 - si.m1, sw0 = 26.0 - si.m2;
    const int NC = W.NC;
    const double *__restrict__ tcoef = W.tcoef;
    const double *etab = W.etab;
    {
        const int idx = min(base + lane, p1 - 1);           // clamped: every lane stays in the loop body
        const bool in_range = base + lane < p1;
        int w2, h2;                                         // 0-based patch coordinates, h fastest
        divmod_small(idx, H2, 1.0f / (float)H2, w2, h2);
        const int h = P.off_h + h2 + 1, w = P.off_w + w2 + 1;  // 1-based image coordinates
        const double hh = (double)h, ww = (double)w;
#define LOAD_PIXEL_INPUTS() load_pixel_inputs<MULTI, COHV>(img, P, W.patches, W.bitmaps, W.nbr_idx, W.nv, W.nb0, W.nb1, W.val_off, \
                                                     W.val, W.active_rank, W.my_rank, W.N, W.n, H2, h, w, h2, w2, in_range)
        // ---- the active source ----
        const bool own_geo = in_range && (w2 < W2 - 1);  // 1 <= w2 < W2 (1-based), elbo_objective.jl:349
        if (MODE == 0) {
            const PixelInputs I = LOAD_PIXEL_INPUTS();
            const bool valid = I.valid, dup = I.dup, own = valid && own_geo;
            const double x = I.x, Ebar = I.Ebar, Vbar = I.Vbar, lgx = I.lgx, iota = I.iota, log_iota = I.log_iota;
            const int n_inact = I.n_inact;
            double f0 = 0, f1 = 0;
            if (own) {
                f0 = star_value(tcoef, hh + sh0, ww + sw0);
                f1 = galaxy_value(W.tc, NC, hh - si.m1, ww - si.m2, etab);
            }
            if (valid) {
                const double c0 = si.c0, c1 = si.c1, q0 = si.q0, q1 = si.q1;
                const double A = c0 * f0 + c1 * f1;
                const double E = Ebar + A;
                const double V = Vbar + ((q0 * (f0 * f0) + q1 * (f1 * f1)) - A * A);
                const double iE = 1.0 / E;
                if (!(MULTI && dup)) a[0] += x * (log_iota + (log(E) - V * (0.5 * iE * iE))) - iota * E - lgx;
                a[1] += own ? 1.0 : 0.0;
                a[2] += (double)n_inact;
            }
            return;
        }

        // S: the arithmetic type of everything per pixel outside the component loop -- double, or float in the
        // single-precision mode (the loop's type R)
        typedef R S;
        PixelTermsT<S> T;
        S S0 = 0;
        T.S0d = 0; T.S1x = 0; T.S1y = 0; T.S1xd = 0; T.S1yd = 0;
        T.S2a = 0; T.S2b = 0; T.S2c = 0; T.S2an = 0; T.S2bn = 0; T.S2cn = 0; T.S2ad = 0; T.S2bd = 0; T.S2cd = 0;
        T.S3a = 0; T.S3b = 0; T.S3c = 0; T.S3d = 0; T.S4a = 0; T.S4b = 0; T.S4c = 0; T.S4d = 0; T.S4e = 0;
        // The component loop runs before the pixel's inputs are fetched: the loop needs only the pixel's coordinates,
        // and every double that is not live across it is a register the loop does not have to share (a masked pixel
        // inside the patch costs one wasted evaluation; its record entries are zeroed by the weights below).
        if (own_geo) {
            if constexpr (sizeof(R) == 4) S0 = galaxy_sums_pk<GM>(W.tcr, 8 * (NC / 14), NC, (float)(hh - si.m1), (float)(ww - si.m2), (float)si.dev, T);
            else S0 = galaxy_sums<GM, R>(W.tcr, 8 * (NC / 14), NC, (R)(hh - si.m1), (R)(ww - si.m2), (R)si.dev, etab, T, W.tcx);
        }
        const PixelInputs I = LOAD_PIXEL_INPUTS();
#undef LOAD_PIXEL_INPUTS
        const bool valid = I.valid, dup = I.dup, own = valid && own_geo;
        const S x = (S)I.x, Ebar = (S)I.Ebar, Vbar = (S)I.Vbar, lgx = (S)I.lgx, iota = (S)I.iota, log_iota = (S)I.log_iota;
        const int n_inact = I.n_inact;
        const S c0 = (S)si.c0, c1 = (S)si.c1, q0 = (S)si.q0, q1 = (S)si.q1;
        T.f1 = own ? S0 : (S)0.0;

        // Star: natural bicubic spline value + derivatives with respect to the index, index = h - m + 26
        T.f0 = 0; T.f0g0 = 0; T.f0g1 = 0; T.f0h0 = 0; T.f0h1 = 0; T.f0h2 = 0;
        if (own) {
            const double xh = hh + sh0, xw = ww + sw0;     // (the cell index and the offset inside the cell from fp64 coordinates)
            int ix = (int)floor(xh); ix = ix < 1 ? 1 : (ix > 50 ? 50 : ix);
            int iy = (int)floor(xw); iy = iy < 1 ? 1 : (iy > 50 ? 50 : iy);
            const double *cc = tcoef + (ix - 1) + CEL_COEF * (iy - 1);
            const S fx = (S)(xh - ix), fy = (S)(xw - iy);
            S wx[4], wy[4], dwx[4], ddwx[4], dwy[4], ddwy[4];
            bspline_w(fx, wx); bspline_w(fy, wy);
            bspline_dw(fx, dwx, ddwx); bspline_dw(fy, dwy, ddwy);
            S y = 0, yx = 0, yy = 0, yxx = 0, yxy = 0, yyy = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const double *cb = cc + CEL_COEF * b;
                const S k0 = (S)cb[0], k1 = (S)cb[1], k2 = (S)cb[2], k3 = (S)cb[3];
                const S r = k0 * wx[0] + k1 * wx[1] + k2 * wx[2] + k3 * wx[3];
                const S rx = k0 * dwx[0] + k1 * dwx[1] + k2 * dwx[2] + k3 * dwx[3];
                const S rxx = k0 * ddwx[0] + k1 * ddwx[1] + k2 * ddwx[2] + k3 * ddwx[3];
                y += r * wy[b]; yx += rx * wy[b]; yxx += rxx * wy[b];
                yy += r * dwy[b]; yxy += rx * dwy[b]; yyy += r * ddwy[b];
            }
            // softpluslikeinv and its derivatives; not C2 at 0, branch exactly (fsm_util.jl:222)
            S gv, gp, gpp;
            // (fp64: the table-driven exponential of the component loop -- y <= 0 here -- instead of the library's)
            if (y < 0) { if constexpr (sizeof(S) == 8) gv = (S)1e-3 * (S)exp_nonpos((double)y, etab); else gv = (S)1e-3 * exp_s(y); gp = gv; gpp = gv; }
            else { gv = (S)1e-3 * (y + (S)1.0); gp = (S)1e-3; gpp = (S)0.0; }
            T.f0 = gv;
            const S ym1 = -yx, ym2 = -yy;  // d(index)/dm = -1
            T.f0g0 = gp * ym1; T.f0g1 = gp * ym2;
            T.f0h0 = gpp * ym1 * ym1 + gp * yxx;
            T.f0h1 = gpp * ym1 * ym2 + gp * yxy;
            T.f0h2 = gpp * ym2 * ym2 + gp * yyy;
        }

        // ---- per-pixel term (add_pixel_term!, add_elbo_log_term!) ----
        {
            const S A = c0 * T.f0 + c1 * T.f1;                       // E_G_s.v
            const S B = q0 * (T.f0 * T.f0) + q1 * (T.f1 * T.f1);     // E_G2_s.v
            const S E = valid ? Ebar + A : (S)1.0;                   // E_G.v
            const S V = Vbar + (B - A * A);                          // var_G.v
            S iE, logE;
            if constexpr (sizeof(S) == 8) { iE = (S)rcp_pos((double)E); logE = (S)log_pos((double)E); }   // (E > 0; a non-finite E stays non-finite)
            else { iE = (S)1.0 / E; logE = log_s(E); }
            const S iE2 = iE * iE, iE3 = iE2 * iE;
            T.vterm = (valid && !(MULTI && dup)) ? x * (log_iota + (logE - V * ((S)0.5 * iE2))) - iota * E - lgx : (S)0.0;
            T.cnt_act = own ? (S)1.0 : (S)0.0;
            T.cnt_inact = (S)n_inact;
            // derivative weights are zero unless the active source covers the pixel, which zeroes every
            // derivative entry of the record (all lanes take part in the cross-lane exchange below)
            const S xo = own ? x : (S)0.0, io = own ? iota : (S)0.0;
            const S w1 = xo * (iE + V * iE3) - io;                   // dT/dE
            T.w2 = (S)-0.5 * xo * iE2;                               // dT/dVar
            const S w11 = -xo * (iE2 + (S)3.0 * V * iE2 * iE2);      // d2T/dE2
            T.w12 = xo * iE3;                                        // d2T/dE dVar
            T.alpha = w1 - (S)2.0 * A * T.w2;
            T.beta = w11 - (S)2.0 * T.w2 - (S)4.0 * A * T.w12;
            T.k1 = T.alpha * c1 + (S)2.0 * T.w2 * q1 * T.f1;         // multiplies d2 f1
            T.k0 = T.alpha * c0 + (S)2.0 * T.w2 * q0 * T.f0;         // multiplies d2 f0
            {
                const S t0 = (S)2.0 * T.w12 * q0 * T.f0, t1 = (S)2.0 * T.w12 * q1 * T.f1;
                const S P0 = T.beta * c0 + t0, P1 = T.beta * c1 + t1;      // beta dA + w12 dB = P0 ds + P1 dg
                T.C0s = T.alpha + T.f0 * P0; T.C0g = T.f0 * P1;
                T.C1s = T.f1 * P0; T.C1g = T.alpha + T.f1 * P1;
                const S v0 = T.w12 * (T.f0 * T.f0), v1 = T.w12 * (T.f1 * T.f1), u0 = (S)2.0 * T.w2 * T.f0, u1 = (S)2.0 * T.w2 * T.f1;
                T.Q0s = u0 + v0 * c0; T.Q0g = v0 * c1;                          // 2 w2 f0 ds + w12 f0^2 dA
                T.Q1s = v1 * c0; T.Q1g = u1 + v1 * c1;
                T.Wgg = (S)2.0 * T.w2 * q1 + c1 * (P1 + t1);
                T.Wss = (S)2.0 * T.w2 * q0 + c0 * (P0 + t0);
                T.Wsg = c0 * P1 + c1 * t0;
            }
        }
        double *const slot_s = slot;
        if constexpr (MODE == 3)
            store_entries<0>(T, W.rec + (size_t)(W.tile_off[W.v] + (base >> 6)) * (ACC_N * 64) + lane);
        else if constexpr (GATED) {
            // several wavefronts share the slots and add in turn: the entries are formed BEFORE a wavefront waits for its
            // turn (they fit the registers the component loop no longer needs), so a turn is 65 LDS adds long, not 600 VALU
            // instructions + 65 adds
            S ent[ACC_N];
            form_entries<MODE, 0>(T, ent);
            gate();
            add_entries<MODE, 0>(ent, slot_s);
        } else {
            gate();
            static_assert(sizeof(S) == 8, "the single-precision derivative modes run pixel_iter_px2");
            accum_entries<MODE, 0>(T, slot_s);
        }
    }
}

// ---- single-precision mode, TWO pixels per lane (pixel_kernel<1 | 2, float>) -------------------------------------------------
// The lane's pixels idx = base + lane and base + 64 + lane sit in the two halves of float2 values, so that EVERYTHING per
// pixel -- the component loop, the star spline, the per-pixel term, the 65 record entries -- runs on v_pk_fma_f32 /
// v_pk_mul_f32 / v_pk_add_f32 (round 4 packed two COMPONENTS per instruction: the 41 % of the kernel's VALU instructions
// outside the component loop stayed scalar).  The component records keep their pair-interleaved LDS layout; component c of a
// pair is the low half of every slot, c + 1 the high half, broadcast to both pixels by the instruction's op_sel bits.  The
// two pixels' record entries are added before they go to the chunk's slots: half the LDS adds per pixel.  Same arithmetic per
// pixel as pixel_iter<MODE, float>; sums are formed in another (fixed) order.
template <int MODE, class TT>
__device__ __forceinline__ f2v galaxy_sums_px2(const f2v *tp, int n_dev, int nc, f2v dx, f2v dy, float dev, TT &T) {
    const f2v z = (f2v)(0.0f);
    if constexpr (MODE == 2) {
        // as in galaxy_sums: the six sums that exist f-weighted and d-weighted are accumulated once, d-weighted, per profile
        // type (U0: de Vaucouleurs, U1: exponential, carrying wd's minus sign)
        f2v U0[6] = {z, z, z, z, z, z}, U1[6] = {z, z, z, z, z, z};
        f2v S2a = z, S2b = z, S2c = z, S3a = z, S3b = z, S3c = z, S3d = z, S4a = z, S4b = z, S4c = z, S4d = z, S4e = z;
        // GW (std::true_type): the sums of orders 2 .. 4 are accumulated with the d-weights g = wd e as well (the de Vaucouleurs
        // loop: f = theta_0 g there, applied once to the twelve sums when the loop ends -- three weight products per component
        // instead of five, as in galaxy_sums).  The exponential is v_exp_f32 directly (base 2).
        // The 8 (de Vaucouleurs) / 6 (exponential) prototypes of a run share a PSF component, i.e. the offset xiBar_k: d = x - xi
        // and hd = -log2(e)/2 d are formed once per run, and the exponent is hd' u in two instructions.
        auto comp = [&](auto HI, auto GW, const f2v *k, f2v (&U)[6], f2v d1, f2v d2, f2v hd1, f2v hd2) {
            constexpr bool hi = decltype(HI)::value, gw = decltype(GW)::value;
#define PXB(i) (hi ? k[i].yy : k[i].xx)
            const f2v p11 = PXB(0), p12 = PXB(1), p22 = PXB(2), w0 = PXB(3), wd = PXB(4), nu = PXB(5);
            const f2v m2p12 = PXB(8), m3p11 = PXB(9), m3p12 = PXB(10), m3p22 = PXB(11);
#undef PXB
            const f2v u = p11 * d1 + p12 * d2, v = p12 * d1 + p22 * d2;
            const f2v q2 = hd1 * u + hd2 * v;                                  // log2(e) x (-1/2) d' P d
            const f2v e = {__builtin_amdgcn_exp2f(q2.x), __builtin_amdgcn_exp2f(q2.y)};
            const f2v g = wd * e, gn = g * nu;
            f2v f, fn, fnn;
            if constexpr (gw) { f = g; fn = gn; fnn = gn * nu; }
            else { f = w0 * e; fn = f * nu; fnn = fn * nu; }
            const f2v ha = u * u - p11, hb = u * v - p12, hc = v * v - p22;
            U[0] += g; U[1] += u * g; U[2] += v * g;
            U[3] += ha * gn; U[4] += hb * gn; U[5] += hc * gn;
            S2a += ha * f; S2b += hb * f; S2c += hc * f;
            const f2v h3a = u * (ha - 2.0f * p11), h3b = v * ha + u * m2p12, h3c = u * hc + v * m2p12, h3d = v * (hc - 2.0f * p22);
            S3a += h3a * fn; S3b += h3b * fn; S3c += h3c * fn; S3d += h3d * fn;
            const f2v h4a = u * h3a + ha * m3p11, h4b = v * h3a + ha * m3p12, h4c = u * h3c + (hb * m2p12 - hc * p11),
                      h4d = u * h3d + hc * m3p12, h4e = v * h3d + hc * m3p22;
            S4a += h4a * fnn; S4b += h4b * fnn; S4c += h4c * fnn; S4d += h4d * fnn; S4e += h4e * fnn;
        };
        auto run = [&](int c0, int len, auto GW, f2v (&U)[6]) {
            const f2v *k0 = tp + (PKSLOTS / 2) * c0;
            const f2v d1 = dx - k0[6].xx, d2 = dy - k0[7].xx;
            const f2v hd1 = -0.72134752044448170368f * d1, hd2 = -0.72134752044448170368f * d2;
            for (int c = c0; c < c0 + len; c += 2) {
                const f2v *k = tp + (PKSLOTS / 2) * c;
                comp(std::false_type(), GW, k, U, d1, d2, hd1, hd2); comp(std::true_type(), GW, k, U, d1, d2, hd1, hd2);
            }
        };
        const f2v th0 = (f2v)(dev), th1 = (f2v)(1.0f - dev);
        // (do-while: psf_K >= 1, so each profile type has at least one run -- a `for` makes the compiler keep a zeroed copy of
        // every sum for the path around the loop: 52 register moves per trip of the pixel loop; measured 0.9 % on config 5)
#ifndef PX2_DOWHILE
#define PX2_DOWHILE 1
#endif
#if PX2_DOWHILE
        { int c0 = 0; do { run(c0, 8, std::true_type(), U0); c0 += 8; } while (c0 < n_dev); }
#else
        for (int c0 = 0; c0 < n_dev; c0 += 8) run(c0, 8, std::true_type(), U0);
#endif
        S2a *= th0; S2b *= th0; S2c *= th0; S3a *= th0; S3b *= th0; S3c *= th0; S3d *= th0;
        S4a *= th0; S4b *= th0; S4c *= th0; S4d *= th0; S4e *= th0;
#if PX2_DOWHILE
        { int c0 = n_dev; do { run(c0, 6, std::false_type(), U1); c0 += 6; } while (c0 < nc); }
#else
        for (int c0 = n_dev; c0 < nc; c0 += 6) run(c0, 6, std::false_type(), U1);
#endif
        T.S0d = U0[0] + U1[0]; T.S1xd = U0[1] + U1[1]; T.S1yd = U0[2] + U1[2];
        T.S2ad = U0[3] + U1[3]; T.S2bd = U0[4] + U1[4]; T.S2cd = U0[5] + U1[5];
        T.S1x = th0 * U0[1] - th1 * U1[1]; T.S1y = th0 * U0[2] - th1 * U1[2];
        T.S2an = th0 * U0[3] - th1 * U1[3]; T.S2bn = th0 * U0[4] - th1 * U1[4]; T.S2cn = th0 * U0[5] - th1 * U1[5];
        T.S2a = S2a; T.S2b = S2b; T.S2c = S2c; T.S3a = S3a; T.S3b = S3b; T.S3c = S3c; T.S3d = S3d;
        T.S4a = S4a; T.S4b = S4b; T.S4c = S4c; T.S4d = S4d; T.S4e = S4e;
        return th0 * U0[0] - th1 * U1[0];
    } else {
        f2v S0 = z, S0d = z, S1x = z, S1y = z, S2an = z, S2bn = z, S2cn = z;
        auto comp = [&](auto HI, const f2v *k, f2v d1, f2v d2, f2v hd1, f2v hd2) {
            constexpr bool hi = decltype(HI)::value;
#define PXB(i) (hi ? k[i].yy : k[i].xx)
            const f2v p11 = PXB(0), p12 = PXB(1), p22 = PXB(2), w0 = PXB(3), wd = PXB(4), nu = PXB(5);
#undef PXB
            const f2v u = p11 * d1 + p12 * d2, v = p12 * d1 + p22 * d2;
            const f2v q2 = hd1 * u + hd2 * v;
            const f2v e = {__builtin_amdgcn_exp2f(q2.x), __builtin_amdgcn_exp2f(q2.y)};
            const f2v f = w0 * e, fd = wd * e, fn = f * nu;
            const f2v ha = u * u - p11, hb = u * v - p12, hc = v * v - p22;
            S0 += f; S0d += fd;
            S1x += u * f; S1y += v * f;
            S2an += ha * fn; S2bn += hb * fn; S2cn += hc * fn;
        };
        auto run = [&](int c0, int len) {
            const f2v *k0 = tp + (PKSLOTS / 2) * c0;
            const f2v d1 = dx - k0[6].xx, d2 = dy - k0[7].xx;
            const f2v hd1 = -0.72134752044448170368f * d1, hd2 = -0.72134752044448170368f * d2;
            for (int c = c0; c < c0 + len; c += 2) {
                const f2v *k = tp + (PKSLOTS / 2) * c;
                comp(std::false_type(), k, d1, d2, hd1, hd2); comp(std::true_type(), k, d1, d2, hd1, hd2);
            }
        };
        { int c0 = 0; do { run(c0, 8); c0 += 8; } while (c0 < n_dev); }
        { int c0 = n_dev; do { run(c0, 6); c0 += 6; } while (c0 < nc); }
        T.S0d = S0d; T.S1x = S1x; T.S1y = S1y; T.S2an = S2an; T.S2bn = S2bn; T.S2cn = S2cn;
        return S0;
    }
}

// entries E .. E + 3 of the two pixels of every lane -> the chunk's slots: the pixels' entries are added, then
// accum_entries_rows' fold over the four rows of 16 lanes (17 conflict-free LDS adds per 128 pixels)
template <int MODE, int E, class TT>
__device__ __forceinline__ float entry2_or_zero(const TT &T) {
    constexpr bool hess_only = E > ZV && E < ACC_CNT;
    if constexpr (E >= ACC_N || (MODE == 1 && hess_only) || entry_is_zero<E < ACC_N ? E : 0>()) return 0.0f;
    else { const f2v e = record_entry<E>(T); return e.x + e.y; }
}
template <int MODE, int E, class TT>
__device__ __forceinline__ void accum_entries_rows2(const TT &T, double *__restrict__ slot_row) {
    static_assert(ACC_N % 4 == 0, "entries are added four at a time");
    if constexpr (!group_is_empty<MODE, E>()) {
        const float s01 = fold_rows_pair16(entry2_or_zero<MODE, E>(T), entry2_or_zero<MODE, E + 1>(T));
        const float s23 = fold_rows_pair16(entry2_or_zero<MODE, E + 2>(T), entry2_or_zero<MODE, E + 3>(T));
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(s01), __float_as_uint(s23), false, false);
        const float s = __uint_as_float(r[0]) + __uint_as_float(r[1]);       // row r: entry E + r, slot lane & 15
        __hip_atomic_fetch_add(slot_row + ACC_SLOTS * E, (double)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if constexpr (E + 4 < ACC_N) accum_entries_rows2<MODE, E + 4>(T, slot_row);
}

// one iteration of the single-precision pixel loop: 128 pixels, lanes = the pixels base + lane and base + 64 + lane
// PX2_GUARD 0 drops the two exec-mask guards below (they almost never skip anything, and without them the compiler keeps one
// zeroed copy of the sums fewer: 912 -> 856 VALU per trip): measured 1.7 % SLOWER on config 5 (5.07 against 4.95 ms) -- the
// masked-off lanes of the guarded region are what the kernel gains, not the skipped trips (profiles/r06_fp32_loop_variants.txt)
#ifndef PX2_GUARD
#define PX2_GUARD 1
#endif
template <int MODE>
__device__ __forceinline__ void pixel_iter_px2(const PixWork<float> &W, int base, int p1, int lane, double *__restrict__ slot) {
    static_assert(MODE == 1 || MODE == 2, "derivative modes only");
    typedef f2v S;
    const DevImage &img = *W.img;
    const DevPatch &P = *W.P;
    const SrcImg &si = W.si;
    const int H2 = P.H2, W2 = P.W2, NC = W.NC;
    const double sh0 = 26.0 - si.m1, sw0 = 26.0 - si.m2;
    const float *__restrict__ tcoef = W.tcoef_f;
#define PK2(a, b) ((S){(float)(a), (float)(b)})
    int h[2], w[2], h2[2], w2[2];
    const float rH2 = 1.0f / (float)H2;
    bool in_range[2], own_geo[2];
    double hh[2], ww[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int i = base + 64 * q + lane;
        const int idx = min(i, p1 - 1);                   // clamped: every lane stays in the loop body
        in_range[q] = i < p1;
        divmod_small(idx, H2, rH2, w2[q], h2[q]);         // 0-based patch coordinates, h fastest
        h[q] = P.off_h + h2[q] + 1; w[q] = P.off_w + w2[q] + 1;
        hh[q] = (double)h[q]; ww[q] = (double)w[q];
        own_geo[q] = in_range[q] && (w2[q] < W2 - 1);     // 1 <= w2 < W2 (1-based), elbo_objective.jl:349
    }
    PixelTermsT<S> T;
    const S z = (S)(0.0f);
    S S0 = z;
    T.S0d = z; T.S1x = z; T.S1y = z; T.S1xd = z; T.S1yd = z;
    T.S2a = z; T.S2b = z; T.S2c = z; T.S2an = z; T.S2bn = z; T.S2cn = z; T.S2ad = z; T.S2bd = z; T.S2cd = z;
    T.S3a = z; T.S3b = z; T.S3c = z; T.S3d = z; T.S4a = z; T.S4b = z; T.S4c = z; T.S4d = z; T.S4e = z;
    // the component loop runs before the pixels' inputs are fetched (it needs only their coordinates)
    if (PX2_GUARD ? (own_geo[0] || own_geo[1]) : true)
        S0 = galaxy_sums_px2<MODE>(reinterpret_cast<const f2v *>(W.tcr), 8 * (NC / 14), NC, PK2(hh[0] - si.m1, hh[1] - si.m1),
                                   PK2(ww[0] - si.m2, ww[1] - si.m2), (float)si.dev, T);
    // (one gather loop over the neighbours for both pixels -- their dependent loads issued together -- measured no faster:
    // 5.30 against 5.27 ms on config 5, three wavefronts per SIMD hide the chains)
    const PixelInputs I0 = load_pixel_inputs<false, false>(img, P, W.patches, W.bitmaps, W.nbr_idx, W.nv, W.nb0, W.nb1, W.val_off, W.val,
                                                          W.active_rank, W.my_rank, W.N, W.n, H2, h[0], w[0], h2[0], w2[0], in_range[0]);
    const PixelInputs I1 = load_pixel_inputs<false, false>(img, P, W.patches, W.bitmaps, W.nbr_idx, W.nv, W.nb0, W.nb1, W.val_off, W.val,
                                                          W.active_rank, W.my_rank, W.N, W.n, H2, h[1], w[1], h2[1], w2[1], in_range[1]);
    const bool valid[2] = {I0.valid, I1.valid};
    const bool own[2] = {valid[0] && own_geo[0], valid[1] && own_geo[1]};
    auto sel = [](const bool (&m)[2], S v) -> S { return (S){m[0] ? v.x : 0.0f, m[1] ? v.y : 0.0f}; };
    const S x = PK2(I0.x, I1.x), Ebar = PK2(I0.Ebar, I1.Ebar), Vbar = PK2(I0.Vbar, I1.Vbar), lgx = PK2(I0.lgx, I1.lgx);
    const S iota = PK2(I0.iota, I1.iota), log_iota = PK2(I0.log_iota, I1.log_iota);
    const S c0 = (S)((float)si.c0), c1 = (S)((float)si.c1), q0 = (S)((float)si.q0), q1 = (S)((float)si.q1);
    T.f1 = sel(own, S0);

    // Star: natural bicubic spline value + derivatives with respect to the index, index = h - m + 26
    T.f0 = z; T.f0g0 = z; T.f0g1 = z; T.f0h0 = z; T.f0h1 = z; T.f0h2 = z;
    if (PX2_GUARD ? (own[0] || own[1]) : true) {
        const float *cc[2];
        float fxs[2], fys[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const double xh = hh[q] + sh0, xw = ww[q] + sw0;   // (the cell index and the offset inside the cell from fp64 coordinates)
            int ix = (int)floor(xh); ix = ix < 1 ? 1 : (ix > 50 ? 50 : ix);
            int iy = (int)floor(xw); iy = iy < 1 ? 1 : (iy > 50 ? 50 : iy);
            cc[q] = tcoef + (ix - 1) + CEL_COEF * (iy - 1);
            fxs[q] = (float)(xh - ix); fys[q] = (float)(xw - iy);
        }
        const S fx = PK2(fxs[0], fxs[1]), fy = PK2(fys[0], fys[1]);
        S wx[4], wy[4], dwx[4], ddwx[4], dwy[4], ddwy[4];
        bspline_w(fx, wx); bspline_w(fy, wy);
        bspline_dw(fx, dwx, ddwx); bspline_dw(fy, dwy, ddwy);
        S y = z, yx = z, yy = z, yxx = z, yxy = z, yyy = z;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const float *ca = cc[0] + CEL_COEF * b, *cb = cc[1] + CEL_COEF * b;
            const S k0 = PK2(ca[0], cb[0]), k1 = PK2(ca[1], cb[1]), k2 = PK2(ca[2], cb[2]), k3 = PK2(ca[3], cb[3]);
            const S r = k0 * wx[0] + k1 * wx[1] + k2 * wx[2] + k3 * wx[3];
            const S rx = k0 * dwx[0] + k1 * dwx[1] + k2 * dwx[2] + k3 * dwx[3];
            const S rxx = k0 * ddwx[0] + k1 * ddwx[1] + k2 * ddwx[2] + k3 * ddwx[3];
            y += r * wy[b]; yx += rx * wy[b]; yxx += rxx * wy[b];
            yy += r * dwy[b]; yxy += rx * dwy[b]; yyy += r * ddwy[b];
        }
        // softpluslikeinv and its derivatives; not C2 at 0, branch exactly (fsm_util.jl:222) -- per pixel
        const bool neg[2] = {y.x < 0.0f, y.y < 0.0f};
        const S ey = (S){1e-3f * exp_s(y.x), 1e-3f * exp_s(y.y)}, lin = 1e-3f * (y + 1.0f);
        const S gv = (S){neg[0] ? ey.x : lin.x, neg[1] ? ey.y : lin.y};
        const S gp = (S){neg[0] ? ey.x : 1e-3f, neg[1] ? ey.y : 1e-3f};
        const S gpp = sel(neg, ey);
        const S ym1 = -yx, ym2 = -yy;  // d(index)/dm = -1
        T.f0 = sel(own, gv);
        T.f0g0 = sel(own, gp * ym1); T.f0g1 = sel(own, gp * ym2);
        T.f0h0 = sel(own, gpp * ym1 * ym1 + gp * yxx);
        T.f0h1 = sel(own, gpp * ym1 * ym2 + gp * yxy);
        T.f0h2 = sel(own, gpp * ym2 * ym2 + gp * yyy);
    }

    // ---- per-pixel term (add_pixel_term!, add_elbo_log_term!) ----
    {
        const S A = c0 * T.f0 + c1 * T.f1;                       // E_G_s.v
        const S B = q0 * (T.f0 * T.f0) + q1 * (T.f1 * T.f1);     // E_G2_s.v
        const S EA = Ebar + A;
        const S E = (S){valid[0] ? EA.x : 1.0f, valid[1] ? EA.y : 1.0f};   // E_G.v
        const S V = Vbar + (B - A * A);                          // var_G.v
        // (v_rcp_f32, 1 ulp, instead of the correctly rounded division's ten instructions: the mode's budget is 1e-4)
        const S iE = (S){__builtin_amdgcn_rcpf(E.x), __builtin_amdgcn_rcpf(E.y)}, logE = (S){log_s(E.x), log_s(E.y)};
        const S iE2 = iE * iE, iE3 = iE2 * iE;
        T.vterm = sel(valid, x * (log_iota + (logE - V * (0.5f * iE2))) - iota * E - lgx);
        T.cnt_act = (S){own[0] ? 1.0f : 0.0f, own[1] ? 1.0f : 0.0f};
        T.cnt_inact = PK2(I0.n_inact, I1.n_inact);
        // derivative weights are zero unless the active source covers the pixel, which zeroes every derivative entry
        const S xo = sel(own, x), io = sel(own, iota);
        const S w1 = xo * (iE + V * iE3) - io;                   // dT/dE
        T.w2 = -0.5f * xo * iE2;                                 // dT/dVar
        const S w11 = -xo * (iE2 + 3.0f * V * iE2 * iE2);        // d2T/dE2
        T.w12 = xo * iE3;                                        // d2T/dE dVar
        T.alpha = w1 - 2.0f * A * T.w2;
        T.beta = w11 - 2.0f * T.w2 - 4.0f * A * T.w12;
        T.k1 = T.alpha * c1 + 2.0f * T.w2 * q1 * T.f1;           // multiplies d2 f1
        T.k0 = T.alpha * c0 + 2.0f * T.w2 * q0 * T.f0;           // multiplies d2 f0
        {
            const S t0 = 2.0f * T.w12 * q0 * T.f0, t1 = 2.0f * T.w12 * q1 * T.f1;
            const S P0 = T.beta * c0 + t0, P1 = T.beta * c1 + t1;      // beta dA + w12 dB = P0 ds + P1 dg
            T.C0s = T.alpha + T.f0 * P0; T.C0g = T.f0 * P1;
            T.C1s = T.f1 * P0; T.C1g = T.alpha + T.f1 * P1;
            const S v0 = T.w12 * (T.f0 * T.f0), v1 = T.w12 * (T.f1 * T.f1), u0 = 2.0f * T.w2 * T.f0, u1 = 2.0f * T.w2 * T.f1;
            T.Q0s = u0 + v0 * c0; T.Q0g = v0 * c1;                          // 2 w2 f0 ds + w12 f0^2 dA
            T.Q1s = v1 * c0; T.Q1g = u1 + v1 * c1;
            T.Wgg = 2.0f * T.w2 * q1 + c1 * (P1 + t1);
            T.Wss = 2.0f * T.w2 * q0 + c0 * (P0 + t0);
            T.Wsg = c0 * P1 + c1 * t0;
        }
    }
#undef PK2
    accum_entries_rows2<MODE, 0>(T, slot + ACC_SLOTS * (lane >> 4));
}

// lane e sums the 16 slots of entry e (rotated start: 4-way instead of 64-way bank conflicts; the order of the additions
// is fixed per entry, so the record is reproducible) and hands the sum to store(e, s)
template <int MODE, typename S, class Store>
__device__ __forceinline__ void fold_record_slots(const S *__restrict__ sacc, int lane, Store &&store) {
    for (int e = lane; e < ACC_N; e += 64) {
        if (MODE == 1 && e > ZV && e < ACC_CNT) continue;   // Hessian entries are not produced
        double s = 0.0;                                      // (fp32 slots are widened here: the record is fp64)
#pragma unroll
        for (int k = 0; k < ACC_SLOTS; ++k) s += (double)sacc[e * ACC_SLOTS + ((k + e) & (ACC_SLOTS - 1))];
        store(e, s);
    }
}

// R: arithmetic type of the galaxy component loop (double; float with CELESTE_FLAG_FP32 -- everything
// downstream of the 24 component sums, and all accumulation, stays fp64)
// MULTI: several active sources (celeste_elbo_eval_multi) -- compiled separately so that the production
// instantiation carries none of its per-neighbour bookkeeping
template <int MODE, typename R, bool MULTI = false>
#ifndef PIXEL_F32_WAVES
#define PIXEL_F32_WAVES 3   // (measured, config 5: 5.37 ms with three waves per SIMD and 76 B of scratch outside the pixel loop, 5.72 ms with two and none)
#endif
__global__ void __launch_bounds__(64, sizeof(R) == 4 ? (MODE == 3 ? 3 : PIXEL_F32_WAVES) : PIXEL_WAVES)
pixel_kernel(const DevImage *__restrict__ images, const DevPatch *__restrict__ patches,
             const double *__restrict__ coefs, const uint8_t *__restrict__ bitmaps,
             const SrcImg *__restrict__ srcimg, const Comp *__restrict__ comps,
             const int64_t *__restrict__ nbr_off, const int32_t *__restrict__ nbr_idx,
             const int64_t *__restrict__ val_off, const double2 *__restrict__ val,
             const int32_t *__restrict__ targets, int N, int NC, int CH, int chunk_px, int G,
             double *__restrict__ acc, const int64_t *__restrict__ tile_off, double *__restrict__ rec,
             const int32_t *__restrict__ active_rank, const int2 *__restrict__ items, int M,
             const int32_t *__restrict__ work, const int32_t *__restrict__ work_total,
             const int64_t *__restrict__ nv_base, const int32_t *__restrict__ nbr_vis,
             const int32_t *__restrict__ rec_off, const float *__restrict__ coefs_f) {
    __shared__ double etab[64];
    // work list (work_fill_kernel): groups of up to G chunks that exist, longest first.  The grid is the host's bound on
    // the list's length: exact when it knows the targets, else the chunk count of the n_targets chunk-richest sources --
    // which a list with repeated targets can exceed, hence the stride loop (one trip in every other case).
    const int n_items = *work_total;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    if (item != (int)blockIdx.x) __syncthreads();      // second trip: every lane is done with the LDS tables
    const int wg0 = work[item];  // record index of the group's first chunk, as the lift kernel expects: ((ti * M + j) * CH + ch)
    const int tn = wg0 / CH;
    const int ch0 = wg0 - tn * CH;
    const int ti = tn / M;                     // tn = ti * M + j: target, j-th image it appears in
    const int t = targets[ti];
    int n = tn - ti * M, v = t * N + n;        // items == nullptr: every source is listed in all M = N images, visit = t N + n
    if (items) { const int2 e = items[tn]; v = e.x; n = e.y; }
    if (v < 0) continue;
    const int rec_base = rec_off[tn];
    const DevPatch &P = patches[v];
    const int H2 = P.H2, W2 = P.W2;
    const int npx = H2 * W2;
    if (ch0 * chunk_px >= npx) continue;  // the lift kernel recomputes this predicate
    const int lane = threadIdx.x;
    // Workgroup prologue: the exp table and the target's components (64-byte records) are staged in LDS with
    // one coalesced read each and a single barrier (the scalar data cache cannot hold 8 waves x 1.8 KB per CU:
    // 68 % of per-component s_loads missed to L2).
    // (the fp64 copy is what the value-only mode and the fp64 loop read: the single-precision instantiations with
    // derivatives do not keep it -- 3.5 KB of LDS less per workgroup, which is what lets a third wavefront per SIMD in)
    constexpr bool keep_tc = sizeof(R) == 8 || MODE == 0;
    __shared__ Comp tc[keep_tc ? 14 * CEL_MAXK + 1 : 1];   // (+ 1: the unrolled component loop requests one record past the last)
    __shared__ float tcr_f[sizeof(R) == 4 ? PKSLOTS * 14 * CEL_MAXK : 1];   // PKSLOTS slots of two floats per pair of components
    {
        const double *src = reinterpret_cast<const double *>(comps + (size_t)v * NC);
        double *dst = reinterpret_cast<double *>(tc);
        R *dstf = reinterpret_cast<R *>(tcr_f);
        const double tabv = g_exp2_table[lane];
        for (int i = lane; i < NC * 8; i += 64) {
            const double v = src[i];
            if (keep_tc) dst[i] = v;
            // fp32 loop: pair-interleaved -- field f of component c at slot (c / 2) PKSLOTS + f, half c & 1, followed by
            // comp_extra's multiples of the precision matrix (galaxy_sums_pk)
            if (sizeof(R) == 4) {
                const int c = i >> 3, f = i & 7;
                R *const rec = dstf + (c >> 1) * (2 * PKSLOTS) + (c & 1);
                rec[2 * f] = (R)v;
                if (f == 0) rec[2 * 9] = (R)(-3.0 * v);
                if (f == 1) { rec[2 * 8] = (R)(-2.0 * v); rec[2 * 10] = (R)(-3.0 * v); }
                if (f == 2) rec[2 * 11] = (R)(-3.0 * v);
            }
        }
        etab[lane] = tabv;
        __syncthreads();
    }
    __shared__ double tcx[(MODE == 2 || MODE == 3) && sizeof(R) == 8 ? COMPX * (14 * CEL_MAXK + 1) : 1];
    if constexpr ((MODE == 2 || MODE == 3) && sizeof(R) == 8) {
        if (lane < NC) comp_extra(tc[lane], tcx + COMPX * lane);
        __syncthreads();
    }
    PixWork<R> W;
    W.img = &images[n]; W.P = &P; W.patches = patches; W.bitmaps = bitmaps; W.nbr_idx = nbr_idx;
    W.nb0 = nbr_off[t]; W.nb1 = nbr_off[t + 1];
    // visit lists: the row of this visit in the table of the neighbours' visits
    W.nv = nbr_vis ? nbr_vis + nv_base[t] + (int64_t)(tn - ti * M) * (W.nb1 - W.nb0) : nullptr;
    W.val_off = val_off; W.val = val; W.active_rank = active_rank;
    // several active sources (celeste_elbo_eval_multi): a pixel of two active patches is visited by the earlier
    // one only (elbo_objective.jl:430-470) -- its value term and inactive-source count are dropped here
    W.my_rank = MULTI ? active_rank[t] : 0;
    W.N = N; W.n = n; W.NC = NC; W.v = v;
    W.si = srcimg[v];
    W.tc = tc; W.tcx = tcx;
    W.tcr = sizeof(R) == 4 ? reinterpret_cast<const CompR<R> *>(tcr_f) : reinterpret_cast<const CompR<R> *>(tc);
    W.etab = etab;
    W.tcoef = coefs + (size_t)(CELESTE_MUTANT == 3 ? 0 : P.stamp) * (CEL_COEF * CEL_COEF);
    W.tcoef_f = coefs_f + (size_t)(CELESTE_MUTANT == 3 ? 0 : P.stamp) * (CEL_COEF * CEL_COEF);
    W.tile_off = tile_off; W.rec = rec;
    // the chunk's record, 16 slots per entry (accum_entries); MODE 0 keeps its three sums in registers
    __shared__ double sacc[MODE == 0 || MODE == 3 ? 1 : ACC_N * ACC_SLOTS];
    double *const slot = sacc + (lane & (ACC_SLOTS - 1));
    for (int ch = ch0; ch < ch0 + G && ch * chunk_px < npx; ++ch) {   // every chunk of the group writes its own record
    const int p0 = ch * chunk_px;
    const int p1 = min(npx, p0 + chunk_px);
    const int wg = rec_base + ch;            // the visit's records are consecutive (rec_off: prefix sum of the chunk counts)
    if constexpr (MODE == 1 || MODE == 2) {
#pragma unroll
        for (int i = 0; i < ACC_N * ACC_SLOTS / 64; ++i) sacc[lane + 64 * i] = 0.0;
        __syncthreads();
    }
    double a[3] = {0.0, 0.0, 0.0};

    if constexpr (sizeof(R) == 4 && (MODE == 1 || MODE == 2) && !MULTI) {
        for (int base = p0; base < p1; base += 128) pixel_iter_px2<MODE>(W, base, p1, lane, slot);
    } else
    for (int base = p0; base < p1; base += 64) pixel_iter<MODE, R, MULTI>(W, base, p1, lane, slot, a, []() {});
    if (MODE == 3) continue;

    // ---- one 68-double record per (target, image, chunk) ----
    double *__restrict__ out = acc + (size_t)wg * ACC_N;
    if (MODE == 0) {
        const double s0 = wave_sum(a[0]), s1 = wave_sum(a[1]), s2 = wave_sum(a[2]);
        if (lane == 0) { out[0] = s0; out[ACC_CNT] = s1; out[ACC_CNT + 1] = s2; }
        continue;
    }
    __syncthreads();
    fold_record_slots<MODE>(sacc, lane, [&](int e, double s) { out[e] = s; });
    __syncthreads();   // the slots are zeroed again for the next chunk of the group
    }
    }   // work item
}

// ---------------------------------------------------------------------------------------------
// record_sum_kernel (split variant): the per-patch sum of the per-pixel records -- the reference's
// accumulation of add_pixel_term! results into elbo_vars.elbo (elbo_objective.jl:330-392, 452-466) as a pure
// HBM-streaming segmented sum.  One workgroup of 128 threads per (target, image); a 64-pixel tile is
// 68 rows x 512 B; thread t reads the double2 at row t/32 + 4k, column pair t%32 (16 B per lane, each wave
// instruction covers two whole rows = 1 KiB contiguous), 17 independent loads per tile.
// ---------------------------------------------------------------------------------------------
typedef double d2v __attribute__((ext_vector_type(2)));
#define RSUM_NT 128
#define RSUM_K (ACC_N * 32 / RSUM_NT)  // 17
__global__ void __launch_bounds__(RSUM_NT)
record_sum_kernel(const DevPatch *__restrict__ patches, const int32_t *__restrict__ targets,
                  const int64_t *__restrict__ tile_off, const double2 *__restrict__ rec,
                  const int2 *__restrict__ items, int N, int M, int RCH, int sum_tiles,
                  double *__restrict__ acc, const int32_t *__restrict__ work, const int32_t *__restrict__ work_total) {
    // part index is the slow grid axis: every patch's first part is launched before any second part.  When a part
    // is a chunk of the pixel kernel (sum_tiles * 64 == chunk_px) the grid runs over that kernel's work list.
    int part, tn;
    if (work) {
        if ((int)blockIdx.x >= *work_total) return;
        const int wg = work[blockIdx.x];
        tn = wg / RCH; part = wg - tn * RCH;
    } else {
        const int TN = gridDim.x / RCH;
        part = blockIdx.x / TN;
        tn = blockIdx.x - part * TN;
    }
    const int ti = tn / M;
    const int t = targets[ti];
    const int vis = items ? items[tn].x : t * N + (tn - ti * M);   // table entry (visit) of (target, image)
    if (vis < 0) return;
    const DevPatch &P = patches[vis];
    const int npx = P.H2 * P.W2;
    const int tile0 = part * sum_tiles;
    const int ntiles = min(sum_tiles, ((npx + 63) >> 6) - tile0);
    if (ntiles <= 0) return;  // the lift kernel recomputes this predicate (part * sum_tiles * 64 < npx)
    const int tid = threadIdx.x;
    const d2v *__restrict__ src = reinterpret_cast<const d2v *>(rec) +
                                  (size_t)(tile_off[vis] + tile0) * (ACC_N * 32) + tid;
    d2v a[RSUM_K];
#pragma unroll
    for (int k = 0; k < RSUM_K; ++k) a[k] = (d2v)(0.0);
    for (int tile = 0; tile < ntiles; ++tile) {
        d2v v[RSUM_K];
#pragma unroll
        for (int k = 0; k < RSUM_K; ++k) v[k] = __builtin_nontemporal_load(src + k * RSUM_NT);
#pragma unroll
        for (int k = 0; k < RSUM_K; ++k) a[k] += v[k];
        src += ACC_N * 32;
    }
    double *__restrict__ out = acc + ((size_t)tn * RCH + part) * ACC_N;
    const int row0 = tid >> 5;
#pragma unroll
    for (int k = 0; k < RSUM_K; ++k) {
        double s = a[k].x + a[k].y;
        s += __shfl_xor(s, 16); s += __shfl_xor(s, 8); s += __shfl_xor(s, 4); s += __shfl_xor(s, 2); s += __shfl_xor(s, 1);
        if ((tid & 31) == 0) out[row0 + 4 * k] = s;
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

// The 10 x 28 Jacobian of the reduced variables is sparse: parameter p only moves the rows param_rows(p) lists (2, 1,
// 3 or 2 of them).  It is stored compactly, entry a of parameter p at jz_off(p) + a: LIFT_JZ = 58 doubles per image.
#define LIFT_JZ 58
__device__ __forceinline__ int jz_off(int p) { return p < 2 ? 2 * p : (p == 2 ? 4 : (p < 6 ? 5 + 3 * (p - 3) : 14 + 2 * (p - 6))); }
__device__ __forceinline__ void jz_unrank(int slot, int &p, int &a) {
    if (slot < 4) { p = slot >> 1; a = slot & 1; }
    else if (slot == 4) { p = 2; a = 0; }
    else if (slot < 14) { const int q = slot - 5; p = 3 + q / 3; a = q - 3 * (p - 3); }
    else { const int q = slot - 14; p = 6 + (q >> 1); a = q & 1; }
}

// index of canonical parameter p inside its type's brightness vector (bids order), -1 for is_star
__device__ inline int bright_slot(int p) {
    if (p < 8) return 0;
    if (p < 10) return 1;
    if (p < 18) return 2 + ((p - 10) & 3);
    if (p < 26) return 6 + ((p - 18) & 3);
    return -1;
}

// exponent coefficients of E_l_a[b, .] (kappa) and E_ll_a[b, .] (lambda) for brightness slot q (bids order), band b
__device__ inline void bright_coef(int q, int b, double &kap, double &lam) {
    kap = 0; lam = 0;
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
}

#ifndef LIFT_NT
#define LIFT_NT 8    // images lifted concurrently per pass
#endif
#define LIFT_NP 28   // parameters with likelihood derivatives (the k block, 28..43, only enters the KL)

// Static tables of the lift (built at compile time): which record entries and Jacobian slots every gradient / Hessian
// entry combines, so that the passes below are straight-line code per entry instead of un-ranking k with a square root
// and walking param_rows / hidx per term.
#define LIFT_JZP (3 * LIFT_NP)          // padded Jacobian: 3 slots per parameter, unused ones hold 0
#define LIFT_REC_ZERO ACC_N             // index of the zero entry appended to every per-image record (padding terms)
struct LiftTables {
    unsigned char p1[LIFT_NP * (LIFT_NP + 1) / 2], p2[LIFT_NP * (LIFT_NP + 1) / 2];   // pair k = p2 (p2 + 1) / 2 + p1
    unsigned char rec[LIFT_NP * (LIFT_NP + 1) / 2][9];   // record entry (Hessian of the reduced variables) of (row a of p1, row c of p2)
    unsigned char code[LIFT_NP * (LIFT_NP + 1) / 2];     // second-derivative term: 0 none, 1 shape x shape, 2 brightness of one type
    unsigned short order[CELESTE_HP];                    // assembly order of the 990 upper-triangle entries: no KL term first, then by KL branch
    unsigned char ap1[CELESTE_HP], ap2[CELESTE_HP], akl[CELESTE_HP];   // entry k = p2 (p2 + 1) / 2 + p1: its parameters, KL branch id (0 = no KL term)
    // the same, one load per entry: pair k -> {rec 0..3, rec 4..7, rec 8 | p1 << 8 | p2 << 16 | code << 24, 0};
    // assembly position kk -> k | p1 << 10 | p2 << 16 | KL branch << 22
    unsigned int pdesc[LIFT_NP * (LIFT_NP + 1) / 2][4];
    unsigned int adesc[CELESTE_HP];
};
constexpr void lift_param_rows(int p, int &start, int &cnt, int &stride, int &cls) {
    if (p < 2) { start = 4; cnt = 2; stride = 1; cls = 0; }
    else if (p == 2) { start = 6; cnt = 1; stride = 1; cls = 1; }
    else if (p < 6) { start = 7; cnt = 3; stride = 1; cls = 2; }
    else if (p < 28) {
        int i = 0;
        if (p < 10) i = (p - 6) & 1;
        else if (p < 26) i = ((p - 10) >> 2) & 1;
        else i = p - 26;
        start = i; cnt = 2; stride = 2; cls = 3 + i;
    } else { start = 0; cnt = 0; stride = 1; cls = 5; }
}
constexpr int lift_kl_branch(int p1, int p2) {   // which return statement of kl_hess (p1 <= p2) is taken; 0: the entry has no KL term
    if (p1 == 5 && p2 == 5) return 1;
    if (p1 < 6 || p2 < 6) return 0;
    auto type_of = [](int p) { return p < 10 ? ((p - 6) & 1) : (p < 26 ? (((p - 10) >> 2) & 1) : (p < 28 ? p - 26 : ((p - 28) >> 3))); };
    auto cls = [](int p) { return p < 8 ? 0 : (p < 10 ? 1 : (p < 18 ? 2 : (p < 26 ? 3 : (p < 28 ? 4 : 5)))); };
    if (type_of(p1) != type_of(p2)) return 0;
    const int c1 = cls(p1), c2 = cls(p2);
    const int q1 = (c1 == 2 || c1 == 3) ? ((p1 - 10) & 3) : (c1 == 5 ? ((p1 - 28) & 7) : 0);
    const int q2 = (c2 == 2 || c2 == 3) ? ((p2 - 10) & 3) : (c2 == 5 ? ((p2 - 28) & 7) : 0);
    if (c2 == 4) return c1 == 4 ? 2 : (c1 == 0 ? 3 : (c1 == 1 ? 4 : 5));
    if (c2 == 5) return c1 == 5 ? (q1 == q2 ? 6 : 0) : (c1 == 4 ? 7 : (c1 == 2 ? 8 : (c1 == 3 ? 9 : 0)));
    if (c1 == 0 && c2 == 0) return 10;
    if (c1 == 1 && c2 == 1) return 11;
    if (c1 == 3 && c2 == 3) return q1 == q2 ? 12 : 0;
    if (c1 == 2 && c2 == 2) return 13;
    return 0;
}
constexpr LiftTables make_lift_tables() {
    LiftTables T = {};
    for (int p2 = 0; p2 < LIFT_NP; ++p2)
        for (int p1 = 0; p1 <= p2; ++p1) {
            const int k = p2 * (p2 + 1) / 2 + p1;
            T.p1[k] = (unsigned char)p1; T.p2[k] = (unsigned char)p2;
            int st1 = 0, cn1 = 0, sd1 = 0, cls1 = 0, st2 = 0, cn2 = 0, sd2 = 0, cls2 = 0;
            lift_param_rows(p1, st1, cn1, sd1, cls1);
            lift_param_rows(p2, st2, cn2, sd2, cls2);
            for (int a = 0; a < 3; ++a)
                for (int c = 0; c < 3; ++c) {
                    int idx = LIFT_REC_ZERO;
                    if (a < cn1 && c < cn2) {
                        const int r1 = st1 + a * sd1, r2 = st2 + c * sd2;
                        idx = hidx(r1 < r2 ? r1 : r2, r1 < r2 ? r2 : r1);
                    }
                    T.rec[k][3 * a + c] = (unsigned char)idx;
                }
            T.code[k] = (cls1 == 2 && cls2 == 2) ? 1 : ((cls1 == cls2 && cls1 >= 3 && cls1 <= 4) ? 2 : 0);
        }
    int n = 0;
    for (int br = 0; br <= 13; ++br)
        for (int p2 = 0; p2 < CEL_P; ++p2)
            for (int p1 = 0; p1 <= p2; ++p1)
                if (lift_kl_branch(p1, p2) == br) T.order[n++] = (unsigned short)(p2 * (p2 + 1) / 2 + p1);
    for (int p2 = 0; p2 < CEL_P; ++p2)
        for (int p1 = 0; p1 <= p2; ++p1) {
            const int k = p2 * (p2 + 1) / 2 + p1;
            T.ap1[k] = (unsigned char)p1; T.ap2[k] = (unsigned char)p2; T.akl[k] = (unsigned char)lift_kl_branch(p1, p2);
        }
    for (int k = 0; k < LIFT_NP * (LIFT_NP + 1) / 2; ++k) {
        T.pdesc[k][0] = T.rec[k][0] | (T.rec[k][1] << 8) | (T.rec[k][2] << 16) | ((unsigned)T.rec[k][3] << 24);
        T.pdesc[k][1] = T.rec[k][4] | (T.rec[k][5] << 8) | (T.rec[k][6] << 16) | ((unsigned)T.rec[k][7] << 24);
        T.pdesc[k][2] = T.rec[k][8] | (T.p1[k] << 8) | (T.p2[k] << 16) | ((unsigned)T.code[k] << 24);
        T.pdesc[k][3] = 0;
    }
    for (int kk = 0; kk < CELESTE_HP; ++kk) {
        const int k = T.order[kk];
        T.adesc[kk] = (unsigned)k | ((unsigned)T.ap1[k] << 10) | ((unsigned)T.ap2[k] << 16) | ((unsigned)T.akl[k] << 22);
    }
    return T;
}
__device__ const __attribute__((aligned(16))) LiftTables c_lift = make_lift_tables();

// Shared state of the analytic KL term (subtract_kl, elbo_kl.jl:94-154)
struct KLShared {
    double t[16], m[16], Ld[16][4], ml[16][4];  // per (type, colour component)
    double ta[2], g[2], g_r[2], g_v[2], ck[2], cm[2];
    double rad;                                  // log p(radius) (elbo_kl.jl:130-137)
};

// Hessian entry (p1 <= p2) of subtract_kl; vs = the target's parameters
__device__ inline double kl_hess(const KLShared &K, const PriorDev *prior, const double *vs, int p1, int p2) {
    if (p1 == 5 && p2 == 5) return -1.0 / prior->p.gal_radius_px_var;
    if (p1 < 6 || p2 < 6) return 0.0;
    // type of each parameter
    auto type_of = [](int p) { return p < 10 ? ((p - 6) & 1) : (p < 26 ? (((p - 10) >> 2) & 1) : (p < 28 ? p - 26 : ((p - 28) >> 3))); };
    const int i = type_of(p1);
    if (type_of(p2) != i) return 0.0;
    const double a = vs[26 + i];
    // classes: 0 r, 1 v, 2 mu_c, 3 lambda_c, 4 a, 5 k_d
    auto cls = [](int p) { return p < 8 ? 0 : (p < 10 ? 1 : (p < 18 ? 2 : (p < 26 ? 3 : (p < 28 ? 4 : 5)))); };
    const int c1 = cls(p1), c2 = cls(p2);
    const int q1 = (c1 == 2 || c1 == 3) ? ((p1 - 10) & 3) : (c1 == 5 ? ((p1 - 28) & 7) : 0);
    const int q2 = (c2 == 2 || c2 == 3) ? ((p2 - 10) & 3) : (c2 == 5 ? ((p2 - 28) & 7) : 0);
    if (c2 == 4) {  // (x, a_i)
        if (c1 == 4) return -1.0 / a;
        if (c1 == 0) return -K.g_r[i];
        if (c1 == 1) return -K.g_v[i];
        double s = 0;
        for (int d = 0; d < 8; ++d) s += vs[28 + 8 * i + d] * (c1 == 2 ? K.Ld[8 * i + d][q1] : -K.ml[8 * i + d][q1]);
        return s;
    }
    if (c2 == 5) {  // (x, k_id)
        const int d = q2;
        if (c1 == 5) return q1 == q2 ? -a / vs[p2] : 0.0;
        if (c1 == 4) return -(K.t[8 * i + d] + 1.0 + K.m[8 * i + d]);
        if (c1 == 2) return a * K.Ld[8 * i + d][q1];
        if (c1 == 3) return -a * K.ml[8 * i + d][q1];
        return 0.0;
    }
    if (c1 == 0 && c2 == 0) return -a / prior->p.flux_var[i];
    if (c1 == 1 && c2 == 1) return -a * 0.5 / (vs[p1] * vs[p1]);
    if (c1 == 3 && c2 == 3) {
        if (q1 != q2) return 0.0;
        double sk = 0;
        for (int d = 0; d < 8; ++d) sk += vs[28 + 8 * i + d];
        return -a * sk * 0.5 / (vs[p1] * vs[p1]);
    }
    if (c1 == 2 && c2 == 2) {
        double s = 0;
        for (int d = 0; d < 8; ++d) s += vs[28 + 8 * i + d] * prior->inv_cov[i][d][q1 + 4 * q2];
        return -a * s;
    }
    return 0.0;
}

__device__ inline double kl_grad(const KLShared &K, const PriorDev *prior, const double *vs, int p) {
    if (p == 5) return -(vs[5] - prior->p.gal_radius_px_mean) / prior->p.gal_radius_px_var;
    if (p < 6) return 0.0;
    const int i = p < 10 ? ((p - 6) & 1) : (p < 26 ? (((p - 10) >> 2) & 1) : (p < 28 ? p - 26 : ((p - 28) >> 3)));
    const double a = vs[26 + i];
    if (p < 8) return -a * K.g_r[i];
    if (p < 10) return -a * K.g_v[i];
    if (p < 26) {
        const int c = (p - 10) & 3;
        double s = 0;
        for (int d = 0; d < 8; ++d) s += vs[28 + 8 * i + d] * (p < 18 ? K.Ld[8 * i + d][c] : -K.ml[8 * i + d][c]);
        return a * s;
    }
    if (p < 28) return -(K.ta[i] + 1.0) - (K.ck[i] + K.g[i] + K.cm[i]);
    const int d = (p - 28) & 7;
    return -a * (K.t[8 * i + d] + 1.0 + K.m[8 * i + d]);
}

#ifdef LIFT_TIMING   // debug builds (tools/variants): shader clocks per phase of the lift kernel, thread 0
__device__ unsigned long long g_lift_clk[16];
#define LIFT_TICK(k) do { const long long now__ = clock64(); if (threadIdx.x == 0) atomicAdd(&g_lift_clk[k], (unsigned long long)(now__ - ltick__)); ltick__ = clock64(); } while (0)
#define LIFT_TICK_DECL long long ltick__ = clock64()
#else
#define LIFT_TICK(k) do { } while (0)
#define LIFT_TICK_DECL do { } while (0)
#endif

// LDS of the lift (17.7 KB)
struct LiftShared {
    double sh_h[LIFT_NP * LIFT_NP];         // likelihood Hessian, upper triangle, params < 28
    double sh_d[LIFT_NP];
    double s_vs[CEL_P], s_jsh[9], s_tsh[27];
    double s_rec[LIFT_NT][ACC_N + 1];       // per-image records; entry ACC_N = 0 (LIFT_REC_ZERO)
    double s_jz[LIFT_NT][LIFT_JZP];         // Jacobians of the reduced variables, 3 slots per parameter (unused: 0)
    double s_J[LIFT_NT][4];                 // the patches' pixel-per-world Jacobians
    double klv;                             // the KL value
    double s_kap[LIFT_NT][10], s_lam[LIFT_NT][10], s_El[LIFT_NT][2], s_Ell[LIFT_NT][2];
    KLShared K;
    double sh_v, sh_cnt[2];
    int sh_bad, own_finite, nbr_finite;
};

// The lift of ONE target by the calling workgroup (>= 256 threads): chunk records -> value, 44-gradient, 44 x 44 Hessian
// at o_v / o_d / o_h (global memory in lift_kernel, LDS in the fused optimiser kernel), KL and status included.
// COH: the target's parameters were written by another workgroup of this launch (ldc), and its SrcGeo is formed here
// from the parameters instead of being read from the per-batch table `geo`.  COH_ACC: so were its chunk records
// (eval_fused_kernel: records from this launch, parameters from before it).
template <bool COH, bool COH_ACC = COH>
__device__ __forceinline__ void lift_target(LiftShared &L, const int tid, int ti, int t, const double *__restrict__ vp,
        const DevImage *__restrict__ images, const DevPatch *__restrict__ patches, const SrcGeo *__restrict__ geo,
        const int64_t *__restrict__ nbr_off, const int32_t *__restrict__ nbr_idx, const double *__restrict__ acc,
        const PriorDev *__restrict__ prior, const int32_t *__restrict__ vis_off, const int32_t *__restrict__ vis_img,
        int N, int M, int CH, int chunk_px, uint32_t flags, double *__restrict__ o_v, double *__restrict__ o_d,
        double *__restrict__ o_h, int64_t *__restrict__ o_cnt, int32_t *__restrict__ o_status,
        const double *__restrict__ lg_sum, const int32_t *__restrict__ rec_off) {
    // lg_sum (per visit, patch_lgamma_kernel; nullptr: the records already hold the term, several active sources)
    LIFT_TICK_DECL;
    double (&sh_h)[LIFT_NP * LIFT_NP] = L.sh_h;
    double (&sh_d)[LIFT_NP] = L.sh_d;
    double (&s_vs)[CEL_P] = L.s_vs, (&s_jsh)[9] = L.s_jsh, (&s_tsh)[27] = L.s_tsh;
    double (&s_rec)[LIFT_NT][ACC_N + 1] = L.s_rec;
    double (&s_jz)[LIFT_NT][LIFT_JZP] = L.s_jz;
    double (&s_kap)[LIFT_NT][10] = L.s_kap, (&s_lam)[LIFT_NT][10] = L.s_lam, (&s_El)[LIFT_NT][2] = L.s_El, (&s_Ell)[LIFT_NT][2] = L.s_Ell;
    KLShared &K = L.K;
    double &sh_v = L.sh_v;
    double (&sh_cnt)[2] = L.sh_cnt;
    int &sh_bad = L.sh_bad;

    const int nthr = blockDim.x;
    const bool want_grad = (flags & (CELESTE_FLAG_GRAD | CELESTE_FLAG_HESS)) != 0;
    const bool want_hess = (flags & CELESTE_FLAG_HESS) != 0;
    const bool want_kl = (flags & CELESTE_FLAG_KL) != 0;

    const int vo = vis_off[t], n_vis = vis_off[t + 1] - vo;   // (first use far below: the loads are in flight meanwhile)
    if (tid < CEL_P) s_vs[tid] = ldc<COH>(vp + (size_t)t * CEL_P + tid);
    if constexpr (!COH) {
        if (tid >= CEL_P && tid < CEL_P + 9) s_jsh[tid - CEL_P] = geo[t].jsh[tid - CEL_P];
        else if (tid >= CEL_P + 9 && tid < CEL_P + 36) s_tsh[tid - CEL_P - 9] = geo[t].tsh[tid - CEL_P - 9];
    }
    for (int k = tid; k < LIFT_NP * LIFT_NP; k += nthr) sh_h[k] = 0.0;
    if (tid < LIFT_NP) sh_d[tid] = 0.0;
    if (tid == 0) { sh_v = 0.0; sh_cnt[0] = 0.0; sh_cnt[1] = 0.0; sh_bad = 0; }
    __syncthreads();
    const double *vs = s_vs;
    // One-thread and few-thread pieces run on threads of the fourth wavefront (192..255) while the other wavefronts are in
    // the first pass; their results are read after that pass's barrier (the Jacobians need the shape derivatives) or at
    // the very end (KL value, finiteness).  lift_wave_sync: LDS hand-over inside one wavefront.
    auto lift_wave_sync = []() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    // The finiteness flags of the target and its neighbours (elbo_objective.jl:487) are three dependent global loads -- offsets,
    // neighbour ids, flags.  Thread 0 used to walk them at the very end of the lift, alone, with the whole workgroup (and, in the
    // optimiser, the target's next Newton step) waiting behind: a spare thread of the third wavefront fetches them now, under
    // the first pass.
    if (tid == 190) {
        int fin = 1;
        if constexpr (!COH) fin = geo[t].finite;
        for (int64_t q = nbr_off[t]; q < nbr_off[t + 1]; ++q) fin &= ldc<COH>(&geo[nbr_idx[q]].finite);
        L.nbr_finite = fin;
    }
    if constexpr (COH) {
        // the target moved since the batch's tables were made: its shape derivatives from its current parameters
        // (source_geo_values, the function setup_thread fills the table with)
        if (tid == 191) {   // (third wavefront; the fourth carries the KL pieces)
            SrcGeo own_geo;
            source_geo_values(s_vs, &own_geo);
            L.own_finite = own_geo.finite;
#pragma unroll
            for (int q = 0; q < 9; ++q) s_jsh[q] = own_geo.jsh[q];
#pragma unroll
            for (int q = 0; q < 27; ++q) s_tsh[q] = own_geo.tsh[q];
        }
    }

    // KL pieces: the fourth wavefront, concurrent with the first lift pass.  Every logarithm and every division the pieces need
    // is evaluated in ONE uniform call with a lane-specific argument (round 4: sixteen threads ran six logarithms and four
    // divisions each, one after the other, then two more threads three logarithms -- 3.7 k cycles of the pass the other
    // wavefronts waited for at its barrier); the pieces are then assembled from lane exchanges with the same expressions in
    // the same order: same bits.
    //   lane  0..15  log k[q]            16..31  log prior.k[q]        32..39  log lam[j], 1 / lam[j]   (j = 4 i + c)
    //        40..41  log a[i], (var1 + (mu1 - mu2)^2) / var2           42..43  log prior.is_star[i], (mu1 - mu2) / var2
    //        44..45  log var2[i], -1 / var1                            46..47  log var1[i], 1 / var2
    //        48      (x - mu)^2 / s2                                   49      log s2
    if (want_kl && tid >= 192 && tid < 256) {
        const int L = tid - 192;
        const celeste_prior_t &pr = prior->p;
        const int i2 = L & 1;
        const double mu1 = vs[6 + i2], var1 = vs[8 + i2], mu2 = pr.flux_mean[i2], var2 = pr.flux_var[i2];
        const double rx = vs[5], rmu = pr.gal_radius_px_mean, rs2 = pr.gal_radius_px_var;
        double lx = 1.0, num = 1.0, den = 1.0;
        if (L < 16) lx = vs[28 + L];
        else if (L < 32) lx = (&pr.k[0][0])[L - 16];
        else if (L < 40) { lx = vs[18 + (L - 32)]; den = lx; }
        else if (L < 42) { lx = vs[26 + i2]; num = var1 + (mu1 - mu2) * (mu1 - mu2); den = var2; }
        else if (L < 44) { lx = pr.is_star[i2]; num = mu1 - mu2; den = var2; }
        else if (L < 46) { lx = var2; num = -1.0; den = var1; }
        else if (L < 48) { lx = var1; den = var2; }
        else if (L == 48) { num = (rx - rmu) * (rx - rmu); den = rs2; }
        else if (L == 49) lx = rs2;
        const double lg = log(lx), dv = num / den;
        const int q = L & 15, qi = q >> 3;
        const double lgpk = __shfl(lg, 16 + q, 64);
        double l4[4], r4[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) { l4[c] = __shfl(lg, 32 + 4 * qi + c, 64); r4[c] = __shfl(dv, 32 + 4 * qi + c, 64); }
        const double lg_pa = __shfl(lg, 42 + i2, 64), lg_v2 = __shfl(lg, 44 + i2, 64), lg_v1 = __shfl(lg, 46 + i2, 64);
        const double dv_r = __shfl(dv, 42 + i2, 64), dv_m = __shfl(dv, 44 + i2, 64), dv_p = __shfl(dv, 46 + i2, 64);
        const double lg_s2 = __shfl(lg, 49, 64);
        if (L < 16) {
            const int i = qi, d = q & 7;
            K.t[q] = lg - lgpk;
            double diff[4], tr = 0, sl = 0, quad = 0;
            for (int c = 0; c < 4; ++c) diff[c] = pr.color_mean[i][d][c] - vs[10 + 4 * i + c];
            for (int c = 0; c < 4; ++c) {
                const double lam = vs[18 + 4 * i + c];
                tr += prior->inv_cov[i][d][c + 4 * c] * lam; sl += l4[c];
            }
            for (int r = 0; r < 4; ++r) {
                double Ld = 0;
                for (int c = 0; c < 4; ++c) Ld += prior->inv_cov[i][d][r + 4 * c] * diff[c];
                K.Ld[q][r] = Ld; quad += diff[r] * Ld;
                K.ml[q][r] = 0.5 * (prior->inv_cov[i][d][r + 4 * r] - r4[r]);
            }
            K.m[q] = 0.5 * ((tr - 4.0) + quad + (prior->logdet[i][d] - sl));
        } else if (L == 40 || L == 41) {
            K.ta[i2] = lg - lg_pa;
            K.g[i2] = .5 * (lg_v2 - lg_v1 + dv - 1.0);
            K.g_r[i2] = dv_r;
            K.g_v[i2] = .5 * (dv_m + dv_p);
        } else if (L == 48) {
            // log p(radius) (elbo_kl.jl:130-137)
            K.rad = -0.5 * (log(2.0 * M_PI) + lg_s2 + dv);
        }
    }
    // ---- KL value (elbo_kl.jl:140-154): by the same wavefront, while the others are in the second pass ----
    auto kl_value = [&]() {
        if (want_kl && tid >= 192 && tid < 256) {
            lift_wave_sync();
            if (tid >= 238 && tid < 240) {
                const int i = tid - 238;
                double ck = 0, cm = 0;
                for (int d = 0; d < 8; ++d) { const double k = vs[28 + 8 * i + d]; ck += k * K.t[8 * i + d]; cm += k * K.m[8 * i + d]; }
                K.ck[i] = ck; K.cm[i] = cm;
            }
            lift_wave_sync();
            if (tid == 238) {
                double kl = 0;
                for (int i = 0; i < 2; ++i) kl -= vs[26 + i] * (K.ta[i] + K.ck[i] + K.g[i] + K.cm[i]);
                kl += K.rad;
                L.klv = kl;
            }
        }
    };

    LIFT_TICK(0);
    // the images the target appears in (its visit list), LIFT_NT at a time
    bool kl_done = false;
    for (int n0 = 0; n0 < n_vis; n0 += LIFT_NT) {
        const int nt = min(LIFT_NT, n_vis - n0);
        // (the descriptor of this thread's first Hessian entry: in flight during the first two passes)
        const int n_pairs = LIFT_NP * (LIFT_NP + 1) / 2;
        uint4 pd_next = *reinterpret_cast<const uint4 *>(c_lift.pdesc[tid < n_pairs ? tid : 0]);
        // pass 1: chunk records -> per-image record; brightness moments and exponent coefficients
        if (tid >= 200 && tid < 200 + 4 * nt) { const int q = tid - 200; L.s_J[q >> 2][q & 3] = patches[vo + n0 + (q >> 2)].J[q & 3]; }
        for (int k = tid; k < nt * ACC_N; k += nthr) {
            const int i = k / ACC_N, e = k - i * ACC_N;
            const DevPatch &P = patches[vo + n0 + i];   // tables are indexed by visit
            const int npx = P.H2 * P.W2;
            double s = 0.0;
            const double *const r0 = acc + (rec_off ? (size_t)rec_off[ti * M + n0 + i] : (size_t)(ti * M + n0 + i) * CH) * ACC_N + e;
            // four loads in flight at a time (COH_ACC: past the L1), added in chunk order -- the plain `if (exists) s += r0[..]`
            // loop waited for every chunk's load before it asked for the next
            for (int c0 = 0; c0 * chunk_px < npx && c0 < CH; c0 += 4) {
                double v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bool ex = c0 + q < CH && (c0 + q) * chunk_px < npx;
                    const double *const ptr = r0 + (size_t)(ex ? c0 + q : c0) * ACC_N;
                    v[q] = COH_ACC ? ldc<true>(ptr) : *ptr;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) if (c0 + q < CH && (c0 + q) * chunk_px < npx) s += v[q];
            }
            s_rec[i][e] = s;
        }
        if (tid < nt) s_rec[tid][LIFT_REC_ZERO] = 0.0;
        if (want_grad) {
            // (threads 96 ...: the first 84 have a second record to sum, the last 64 carry the KL pieces)
            for (int k = tid >= 96 ? tid - 96 : tid + nthr - 96; k < nt * 12; k += nthr) {
                const int i = k / 12, q = k - i * 12;
                const int b = images[vis_img[vo + n0 + i]].band - 1;
                if (q < 10) {
                    double kap, lam;
                    bright_coef(q, b, kap, lam);
                    s_kap[i][q] = kap; s_lam[i][q] = lam;
                } else {
                    double El, Ell;
                    brightness(vs, q - 10, b, El, Ell);
                    s_El[i][q - 10] = El; s_Ell[i][q - 10] = Ell;
                }
            }
        }
        __syncthreads();
        LIFT_TICK(1);
        if (!kl_done) { kl_value(); kl_done = true; }
        if (tid == 0) for (int i = 0; i < nt; ++i) {
            sh_v += lg_sum ? s_rec[i][0] - lg_sum[vo + n0 + i] : s_rec[i][0];
            sh_cnt[0] += s_rec[i][ACC_CNT]; sh_cnt[1] += s_rec[i][ACC_CNT + 1];
        }
        if (want_grad) {
            // pass 2: the 10 x 28 Jacobians of the reduced variables, 3 slots per parameter (parameter p moves 2, 1, 3 or 2
            // rows -- param_rows; the unused slots hold 0)
            for (int k = tid; k < nt * LIFT_JZP; k += nthr) {
                const int i = k / LIFT_JZP, slot = k - i * LIFT_JZP;
                const int p = slot / 3, a = slot - 3 * p;
                double v = 0.0;
                if (p < 2) { if (a < 2) v = L.s_J[i][a + 2 * p]; }                        // rows 4, 5: pixel position
                else if (p == 2) { if (a == 0) v = 1.0; }                                 // row 6: gal_frac_dev
                else if (p < 6) v = s_jsh[a + 3 * (p - 3)];                               // rows 7..9: the covariance entries
                else if (a < 2) {                                                         // rows ty (a = 0: c_ty), 2 + ty (a = 1: q_ty)
                    const int ty = p < 10 ? ((p - 6) & 1) : (p < 26 ? (((p - 10) >> 2) & 1) : p - 26);
                    const bool isq = a == 1;
                    const int bs = bright_slot(p);
                    const double Ev = isq ? s_Ell[i][ty] : s_El[i][ty];
                    if (bs < 0) v = Ev;
                    else v = vs[26 + ty] * Ev * (isq ? s_lam[i][bs] : s_kap[i][bs]);
                }
                s_jz[i][slot] = v;
            }
            __syncthreads();
            LIFT_TICK(2);
            // pass 3: gradient and upper-triangle Hessian; every entry is owned by one thread
            for (int k = tid; k < n_pairs + LIFT_NP; k += nthr) {
                const uint4 pd = pd_next;
                if (k + nthr < n_pairs) pd_next = *reinterpret_cast<const uint4 *>(c_lift.pdesc[k + nthr]);
                if (k >= n_pairs) {
                    const int p = k - n_pairs;
                    int st, cn, sd, cls;
                    param_rows(p, st, cn, sd, cls);
                    double s = 0.0;
                    for (int i = 0; i < nt; ++i)
                        for (int a = 0; a < cn; ++a) { const int r = st + a * sd; s += s_jz[i][3 * p + a] * s_rec[i][1 + r]; }
                    sh_d[p] += s;
                    continue;
                }
                if (!want_hess) continue;
                // J1' R J2 over the (up to 3 x 3) rows the two parameters move: the record entries from c_lift.pdesc, terms
                // that do not exist are 0 x 0 (LIFT_REC_ZERO, empty Jacobian slots)
                const int p1 = (pd.z >> 8) & 0xff, p2 = (pd.z >> 16) & 0xff, code = pd.z >> 24;
                int ri[9];
#pragma unroll
                for (int q = 0; q < 4; ++q) { ri[q] = (pd.x >> (8 * q)) & 0xff; ri[4 + q] = (pd.y >> (8 * q)) & 0xff; }
                ri[8] = pd.z & 0xff;
                const int ty = p1 < 10 ? ((p1 - 6) & 1) : (p1 < 26 ? (((p1 - 10) >> 2) & 1) : p1 - 26);   // (code 2 only)
                const int b1 = bright_slot(p1), b2 = bright_slot(p2);
                double s = 0.0;
                for (int i = 0; i < nt; ++i) {
                    const double *rec = s_rec[i];
                    const double *jz = s_jz[i];
                    // (all fifteen LDS reads of the image first: one wait instead of one per term)
                    double rr[9], j1[3];
#pragma unroll
                    for (int q = 0; q < 9; ++q) rr[q] = rec[ri[q]];
#pragma unroll
                    for (int a = 0; a < 3; ++a) j1[a] = jz[3 * p1 + a];
                    const double j20 = jz[3 * p2], j21 = jz[3 * p2 + 1], j22 = jz[3 * p2 + 2];
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        double inner = 0.0;
                        inner += rr[3 * a] * j20;
                        inner += rr[3 * a + 1] * j21;
                        inner += rr[3 * a + 2] * j22;
                        s += j1[a] * inner;
                    }
                    // second derivatives of the reduced variables
                    if (code == 1) {
                        for (int sg = 0; sg < 3; ++sg) s += rec[1 + 7 + sg] * s_tsh[sg + 3 * (p1 - 3) + 9 * (p2 - 3)];
                    } else if (code == 2) {
                        const double gc = rec[1 + ty], gq = rec[1 + 2 + ty];
                        const double El = s_El[i][ty], Ell = s_Ell[i][ty], ai = vs[26 + ty];
                        if (b1 >= 0 && b2 >= 0)
                            s += ai * (gc * El * s_kap[i][b1] * s_kap[i][b2] + gq * Ell * s_lam[i][b1] * s_lam[i][b2]);
                        else if (b1 >= 0) s += gc * El * s_kap[i][b1] + gq * Ell * s_lam[i][b1];
                        else if (b2 >= 0) s += gc * El * s_kap[i][b2] + gq * Ell * s_lam[i][b2];
                    }
                }
                sh_h[p1 + LIFT_NP * p2] += s;
            }
        }
        __syncthreads();
        LIFT_TICK(3);
    }

    if (!kl_done) kl_value();   // (a target may have no visit at all)
    __syncthreads();   // the KL pieces were written by threads of the fourth wavefront
    if (want_kl && tid == 0) sh_v += L.klv;
    LIFT_TICK(4);

    // ---- assemble, check finiteness (elbo_objective.jl:487,490), store exactly symmetric ----
    int bad = 0;
    if (tid == 0 && !isfinite(sh_v)) bad = 1;
    if (want_grad && o_d && tid < CEL_P) {
        double v = tid < LIFT_NP ? sh_d[tid] : 0.0;
        if (want_kl) v += kl_grad(K, prior, vs, tid);
        if (!isfinite(v)) bad = 1;
        o_d[tid] = v;
    }
    if (want_hess && o_h && COH) {
        // (the fused optimiser: o_h is in LDS) every entry (p1 <= p2) of the upper triangle is formed once and stored at
        // (p1, p2) and (p2, p1): half the kl_hess evaluations of the loop below, same values
        // (entries without a KL term first, the others grouped by the branch of kl_hess they take: c_lift.order)
        unsigned ad_next = c_lift.adesc[tid < CELESTE_HP ? tid : 0];
        for (int kk = tid; kk < CELESTE_HP; kk += nthr) {
            const unsigned ad = ad_next;
            if (kk + nthr < CELESTE_HP) ad_next = c_lift.adesc[kk + nthr];
            const int k = ad & 0x3ff, p1 = (ad >> 10) & 0x3f, p2 = (ad >> 16) & 0x3f;
            double v = (p2 < LIFT_NP) ? sh_h[p1 + LIFT_NP * p2] : 0.0;
            if (want_kl && (ad >> 22)) v += kl_hess(K, prior, vs, p1, p2);
            if (!isfinite(v)) bad = 1;
            if (!(flags & CELESTE_FLAG_PACKED_HESS)) { o_h[p1 + CEL_P * p2] = v; o_h[p2 + CEL_P * p1] = v; }
            else o_h[k] = v;   // upper triangle, by columns: (p1, p2) at p2 (p2 + 1) / 2 + p1
        }
    } else if (want_hess && o_h) {
        // (o_h in HBM) Every entry (p1 <= p2) of the upper triangle is formed ONCE, in the order that groups the entries by the
        // branch of kl_hess they take (c_lift.adesc: a wavefront's lanes run the same branch), and parked in LDS -- the
        // per-image records and Jacobians are dead by now, 990 doubles fit over them; then consecutive threads store
        // consecutive entries of the full, exactly symmetric matrix.  (Round 4 formed all 1936 entries in storage order: twice
        // the KL evaluations, lanes of one wavefront in different branches -- a third of the kernel's time.)  Same values.
        static_assert(sizeof(L.s_rec) + sizeof(L.s_jz) >= CELESTE_HP * sizeof(double), "the packed triangle lies over s_rec and s_jz");
        static_assert(offsetof(LiftShared, s_jz) == offsetof(LiftShared, s_rec) + sizeof(L.s_rec), "s_rec and s_jz are adjacent");
        double *const packed = &L.s_rec[0][0];
        unsigned ad_next = c_lift.adesc[tid < CELESTE_HP ? tid : 0];
        for (int kk = tid; kk < CELESTE_HP; kk += nthr) {
            const unsigned ad = ad_next;
            if (kk + nthr < CELESTE_HP) ad_next = c_lift.adesc[kk + nthr];
            const int k = ad & 0x3ff, p1 = (ad >> 10) & 0x3f, p2 = (ad >> 16) & 0x3f;
            double v = (p2 < LIFT_NP) ? sh_h[p1 + LIFT_NP * p2] : 0.0;
            if (want_kl && (ad >> 22)) v += kl_hess(K, prior, vs, p1, p2);
            if (!isfinite(v)) bad = 1;
            packed[k] = v;     // upper triangle, by columns: (p1, p2) at p2 (p2 + 1) / 2 + p1
        }
        __syncthreads();
        if (flags & CELESTE_FLAG_PACKED_HESS) {
            for (int k = tid; k < CELESTE_HP; k += nthr) o_h[k] = packed[k];
        } else {
            for (int k = tid; k < CEL_P * CEL_P; k += nthr) {
                const int c2 = k / CEL_P, c1 = k - c2 * CEL_P;
                const int p1 = c1 < c2 ? c1 : c2, p2 = c1 < c2 ? c2 : c1;
                o_h[k] = packed[p2 * (p2 + 1) / 2 + p1];
            }
        }
    }
    if (bad) atomicOr(&sh_bad, 1);
    __syncthreads();
    if (tid == 0) {
        int st = sh_bad ? CELESTE_ERR_NONFINITE_RESULT : CELESTE_OK;
        const int fin = (COH ? L.own_finite : 1) & L.nbr_finite;
        if (!fin) st = CELESTE_ERR_NONFINITE_INPUT;
        *o_status = st;
        *o_v = sh_v;
        if (o_cnt) { o_cnt[0] = (int64_t)(sh_cnt[0] + 0.5); o_cnt[1] = (int64_t)(sh_cnt[1] + 0.5); }
    }
    LIFT_TICK(5);
#ifdef LIFT_TIMING
    if (tid == 0) atomicAdd(&g_lift_clk[15], 1ull);
#endif
}

// WAVES = 8 per SIMD (64 VGPRs, a 36-byte spill) and 20 KB of LDS: 8 workgroups per CU, so that a 2000-target batch is resident
// in one round (with 5 per CU it ran in two: 76 -> 65 us).  WAVES = 7 (72 VGPRs, an 8-byte spill) for batches that take many rounds
// anyway: the spill is stored by every thread -- 36 B x 256 threads x 30 000 targets = 276 MB of the 724 MB the lift of config 5
// wrote for 476 MB of results (launch_eval picks by batch size).  Same code, same results.
template <int WAVES>
__global__ void __launch_bounds__(512, WAVES)
lift_kernel(const double *__restrict__ vp, const DevImage *__restrict__ images,
            const DevPatch *__restrict__ patches, const SrcGeo *__restrict__ geo,
            const int64_t *__restrict__ nbr_off, const int32_t *__restrict__ nbr_idx,
            const int32_t *__restrict__ targets, const double *__restrict__ acc,
            const PriorDev *__restrict__ prior, const int32_t *__restrict__ vis_off,
            const int32_t *__restrict__ vis_img, int N, int M, int CH, int chunk_px, uint32_t flags,
            double *__restrict__ out_v, double *__restrict__ out_d, double *__restrict__ out_h,
            int64_t *__restrict__ out_cnt, int32_t *__restrict__ out_status, const int32_t *__restrict__ live,
            const double *__restrict__ lg_sum, const int32_t *__restrict__ rec_off, const int32_t *__restrict__ h_pos) {
    if (live && (int)blockIdx.x >= *live) return;
    __shared__ LiftShared L;
    const int ti = blockIdx.x;
    const size_t HS = (flags & CELESTE_FLAG_PACKED_HESS) ? CELESTE_HP : (size_t)CEL_P * CEL_P;
    // h_pos (optional): the Hessian of target ti goes to slot h_pos[ti] of out_h -- a device group's member writes its shard's
    // Hessians to their places in the CALLER's (mapped, page-locked) array
    const size_t hslot = h_pos ? (size_t)h_pos[ti] : (size_t)ti;
    lift_target<false>(L, threadIdx.x, ti, targets[ti], vp, images, patches, geo, nbr_off, nbr_idx, acc, prior, vis_off, vis_img, N, M, CH,
                       chunk_px, flags, out_v + ti, out_d ? out_d + (size_t)ti * CEL_P : nullptr,
                       out_h ? out_h + hslot * HS : nullptr, out_cnt ? out_cnt + 2 * (size_t)ti : nullptr, out_status + ti,
                       lg_sum, rec_off);
}

// ---------------------------------------------------------------------------------------------
// Several active sources (ElboArgs.active_sources, Sa > 1; add_sources_sf!, SensitiveFloats.jl:215-250).
// Gradient columns and diagonal Hessian blocks are the single-active evaluations (every other source value-only);
// the cross block of two active sources a, b collects, over the pixels both cover,
//     d2T/dE2 dE_a dE_b' + d2T/dE dVar (dE_a dVar_b' + dVar_a dE_b')        (d2T/dVar2 = 0)
// first in the reduced variables (cross_kernel: one 10 x 10 record per pair and image), then in the canonical
// parameters (cross_lift_kernel: J_a' M J_b).  A test-level path of the reference: written for clarity, not speed.
// ---------------------------------------------------------------------------------------------
struct FirstOrder { double A, B, gE[ZV], gB[ZV]; };  // E_G_s.v, E_G2_s.v and their gradients in the reduced variables

__device__ inline void source_first_order(const SrcImg &si, const Comp *comps, int NC, const double *tcoef, double hh,
                                          double ww, const double *etab, FirstOrder &F) {
    PixelTerms T;
    T.S0d = 0; T.S1x = 0; T.S1y = 0; T.S2an = 0; T.S2bn = 0; T.S2cn = 0;
    const double f1 = galaxy_sums<1, double>(reinterpret_cast<const CompR<double> *>(comps), 8 * (NC / 14), NC, hh - si.m1,
                                             ww - si.m2, si.dev, etab, T);
    // star: natural bicubic spline value and first derivatives with respect to the position
    const double xh = hh + (26.0 - si.m1), xw = ww + (26.0 - si.m2);
    int ix = (int)floor(xh); ix = ix < 1 ? 1 : (ix > 50 ? 50 : ix);
    int iy = (int)floor(xw); iy = iy < 1 ? 1 : (iy > 50 ? 50 : iy);
    double wx[4], wy[4], dwx[4], ddwx[4], dwy[4], ddwy[4];
    bspline_w(xh - ix, wx); bspline_w(xw - iy, wy);
    bspline_dw(xh - ix, dwx, ddwx); bspline_dw(xw - iy, dwy, ddwy);
    const double *cc = tcoef + (ix - 1) + CEL_COEF * (iy - 1);
    double y = 0, yx = 0, yy = 0;
    for (int b = 0; b < 4; ++b) {
        const double *cb = cc + CEL_COEF * b;
        const double r = cb[0] * wx[0] + cb[1] * wx[1] + cb[2] * wx[2] + cb[3] * wx[3];
        const double rx = cb[0] * dwx[0] + cb[1] * dwx[1] + cb[2] * dwx[2] + cb[3] * dwx[3];
        y += r * wy[b]; yx += rx * wy[b]; yy += r * dwy[b];
    }
    double f0, gp;
    if (y < 0) { f0 = 1e-3 * exp(y); gp = f0; } else { f0 = 1e-3 * (y + 1.0); gp = 1e-3; }
    const double f0g0 = -gp * yx, f0g1 = -gp * yy;   // d(index)/dm = -1
    const double c0 = si.c0, c1 = si.c1, q0 = si.q0, q1 = si.q1;
    F.A = c0 * f0 + c1 * f1;
    F.B = q0 * (f0 * f0) + q1 * (f1 * f1);
    const double gg[6] = {gal_g<0>(T), gal_g<1>(T), gal_g<2>(T), gal_g<3>(T), gal_g<4>(T), gal_g<5>(T)};
    F.gE[0] = f0; F.gE[1] = f1; F.gE[2] = 0; F.gE[3] = 0;
    F.gB[0] = 0; F.gB[1] = 0; F.gB[2] = f0 * f0; F.gB[3] = f1 * f1;
    for (int k = 0; k < 6; ++k) {
        const double sg = k == 0 ? f0g0 : (k == 1 ? f0g1 : 0.0);
        F.gE[4 + k] = c0 * sg + c1 * gg[k];
        F.gB[4 + k] = 2.0 * q0 * f0 * sg + 2.0 * q1 * f1 * gg[k];
    }
}

__global__ void __launch_bounds__(64)
cross_kernel(const DevImage *__restrict__ images, const DevPatch *__restrict__ patches, const double *__restrict__ coefs,
             const uint8_t *__restrict__ bitmaps, const SrcImg *__restrict__ srcimg, const Comp *__restrict__ comps,
             const int64_t *__restrict__ nbr_off, const int32_t *__restrict__ nbr_idx,
             const int64_t *__restrict__ val_off, const double2 *__restrict__ val, const int32_t *__restrict__ pair_a,
             const int32_t *__restrict__ pair_b, int N, int NC, double *__restrict__ out,
             const int32_t *__restrict__ vis_off, const int32_t *__restrict__ vis_img, int dense) {
    __shared__ double etab[64];
    __shared__ double sX[2 * ZV][64], sY[2 * ZV][64];
    const int pair = blockIdx.x / N, n = blockIdx.x - pair * N;
    const int a = pair_a[pair], b = pair_b[pair];
    const int lane = threadIdx.x;
    etab[lane] = g_exp2_table[lane];
    const int va = table_entry(vis_off, vis_img, dense, N, a, n), vb = table_entry(vis_off, vis_img, dense, N, b, n);
    const bool both = va >= 0 && vb >= 0;                  // else: one of them has no patch in this image
    const DevPatch &Pa = patches[both ? va : 0], &Pb = patches[both ? vb : 0];
    const DevImage &img = images[n];
    // pixels both sources light up: rows of both patches, columns of both minus each one's last
    const int h_lo = max(Pa.off_h, Pb.off_h), h_hi = min(Pa.off_h + Pa.H2, Pb.off_h + Pb.H2);
    const int w_lo = max(Pa.off_w, Pb.off_w), w_hi = min(Pa.off_w + Pa.W2 - 1, Pb.off_w + Pb.W2 - 1);
    const int rh = both ? h_hi - h_lo : 0, rw = w_hi - w_lo;
    double acc0 = 0.0, acc1 = 0.0;
    const int e0 = lane, e1 = lane + 64;                   // entries r1 * 10 + r2 owned by this lane
    __syncthreads();
    if (rh > 0 && rw > 0) {
        const SrcImg sa = srcimg[va], sb = srcimg[vb];
        const Comp *ca = comps + (size_t)va * NC, *cb = comps + (size_t)vb * NC;
        const double *ta = coefs + (size_t)Pa.stamp * (CEL_COEF * CEL_COEF), *tb = coefs + (size_t)Pb.stamp * (CEL_COEF * CEL_COEF);
        const int npx = rh * rw;
        for (int base = 0; base < npx; base += 64) {
            const int idx = min(base + lane, npx - 1);
            const int wq = idx / rh, hq = idx - wq * rh;
            const int h = h_lo + hq + 1, w = w_lo + wq + 1;     // 1-based image coordinates
            const size_t gi = (size_t)(h - 1) + (size_t)img.H * (w - 1);
            const float xf = img.pixels[gi];
            bool ok = (base + lane < npx) && !isnan(xf);
            const int ah = h - Pa.off_h, aw = w - Pa.off_w, bh = h - Pb.off_h, bw = w - Pb.off_w;
            if (ok && Pa.bitmap_off >= 0) ok = bitmaps[Pa.bitmap_off + (ah - 1) + (int64_t)Pa.H2 * (aw - 1)] != 0;
            if (ok && Pb.bitmap_off >= 0) ok = bitmaps[Pb.bitmap_off + (bh - 1) + (int64_t)Pb.H2 * (bw - 1)] != 0;
            FirstOrder Fa, Fb;
            source_first_order(sa, ca, NC, ta, (double)h, (double)w, etab, Fa);
            source_first_order(sb, cb, NC, tb, (double)h, (double)w, etab, Fb);
            // totals over every source covering the pixel: a itself + a's neighbours (b among them)
            double E = (double)img.sky[gi] + Fa.A, V = Fa.B - Fa.A * Fa.A;
            for (int64_t q = nbr_off[a]; q < nbr_off[a + 1]; ++q) {
                const int s2 = nbr_idx[q];
                const int v2 = table_entry(vis_off, vis_img, dense, N, s2, n);
                if (v2 < 0) continue;
                const DevPatch &Q = patches[v2];
                const int ph2 = h - Q.off_h, pw2 = w - Q.off_w;
                bool in = ok & (ph2 >= 1) & (ph2 <= Q.H2) & (pw2 >= 1) & (pw2 < Q.W2);
                if (in && Q.bitmap_off >= 0) in = bitmaps[Q.bitmap_off + (ph2 - 1) + (int64_t)Q.H2 * (pw2 - 1)] != 0;
                if (in) {
                    const double2 ev = val[val_off[v2] + (ph2 - 1) + (int64_t)Q.H2 * (pw2 - 1)];
                    E += ev.x; V += ev.y;
                }
            }
            const double x = ok ? (double)xf : 0.0;
            const double iE = 1.0 / E, iE2 = iE * iE;
            const double w11 = -x * (iE2 + 3.0 * V * iE2 * iE2), w12 = x * iE2 * iE;
            for (int r = 0; r < ZV; ++r) {
                const double gVa = Fa.gB[r] - 2.0 * Fa.A * Fa.gE[r], gVb = Fb.gB[r] - 2.0 * Fb.A * Fb.gE[r];
                sX[r][lane] = ok ? w11 * Fa.gE[r] + w12 * gVa : 0.0;
                sX[ZV + r][lane] = ok ? w12 * Fa.gE[r] : 0.0;
                sY[r][lane] = ok ? Fb.gE[r] : 0.0;
                sY[ZV + r][lane] = ok ? gVb : 0.0;
            }
            __syncthreads();
            {
                const int r1 = e0 / ZV, r2 = e0 - r1 * ZV;
                for (int px = 0; px < 64; ++px) acc0 += sX[r1][px] * sY[r2][px] + sX[ZV + r1][px] * sY[ZV + r2][px];
                if (e1 < ZV * ZV) {
                    const int q1 = e1 / ZV, q2 = e1 - q1 * ZV;
                    for (int px = 0; px < 64; ++px) acc1 += sX[q1][px] * sY[q2][px] + sX[ZV + q1][px] * sY[ZV + q2][px];
                }
            }
            __syncthreads();
        }
    }
    double *o = out + (size_t)blockIdx.x * (ZV * ZV);
    o[e0] = acc0;
    if (e1 < ZV * ZV) o[e1] = acc1;
}

// d z_r / d theta_p of one source on one image (the entries lift_kernel tabulates in s_jz)
__device__ inline double zjac_entry(int r, int p, const double *vs, const DevPatch &P, const double *jsh, int band0) {
    if (r >= 4 && r < 6) return p < 2 ? P.J[(r - 4) + 2 * p] : 0.0;
    if (r == 6) return p == 2 ? 1.0 : 0.0;
    if (r >= 7) return (p >= 3 && p < 6) ? jsh[(r - 7) + 3 * (p - 3)] : 0.0;
    const int ty = r & 1;
    const bool isq = r >= 2;
    int st, cn, sd, cls;
    param_rows(p, st, cn, sd, cls);
    if (cls != 3 + ty) return 0.0;
    double El, Ell;
    brightness(vs, ty, band0, El, Ell);
    const double Ev = isq ? Ell : El;
    const int slot = bright_slot(p);
    if (slot < 0) return Ev;
    double kap, lam;
    bright_coef(slot, band0, kap, lam);
    return vs[26 + ty] * Ev * (isq ? lam : kap);
}

// one workgroup per pair: out[pair][p1 + 28 p2] = sum_n sum_{r1, r2} J_a[r1, p1] M_n[r1, r2] J_b[r2, p2]
__global__ void __launch_bounds__(256)
cross_lift_kernel(const double *__restrict__ vp, const DevImage *__restrict__ images,
                  const DevPatch *__restrict__ patches, const SrcGeo *__restrict__ geo,
                  const int32_t *__restrict__ pair_a, const int32_t *__restrict__ pair_b,
                  const double *__restrict__ rec, int N, double *__restrict__ out,
                  const int32_t *__restrict__ vis_off, const int32_t *__restrict__ vis_img, int dense) {
    __shared__ double sJa[ZV * LIFT_NP], sJb[ZV * LIFT_NP], sM[ZV * ZV], sva[CEL_P], svb[CEL_P];
    const int pair = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    const int a = pair_a[pair], b = pair_b[pair];
    if (tid < CEL_P) { sva[tid] = vp[(size_t)a * CEL_P + tid]; svb[tid] = vp[(size_t)b * CEL_P + tid]; }
    double hacc[(LIFT_NP * LIFT_NP + 255) / 256];
    for (int k = 0; k < (LIFT_NP * LIFT_NP + 255) / 256; ++k) hacc[k] = 0.0;
    __syncthreads();
    for (int n = 0; n < N; ++n) {
        const int band0 = images[n].band - 1;
        const int va = table_entry(vis_off, vis_img, dense, N, a, n), vb = table_entry(vis_off, vis_img, dense, N, b, n);
        if (va < 0 || vb < 0) continue;   // (uniform) no common image: the record is zero
        for (int k = tid; k < ZV * LIFT_NP; k += nthr) {
            const int r = k / LIFT_NP, p = k - r * LIFT_NP;
            sJa[k] = zjac_entry(r, p, sva, patches[va], geo[a].jsh, band0);
            sJb[k] = zjac_entry(r, p, svb, patches[vb], geo[b].jsh, band0);
        }
        for (int k = tid; k < ZV * ZV; k += nthr) sM[k] = rec[((size_t)pair * N + n) * (ZV * ZV) + k];
        __syncthreads();
        for (int k = tid, slot = 0; k < LIFT_NP * LIFT_NP; k += nthr, ++slot) {
            const int p2 = k / LIFT_NP, p1 = k - p2 * LIFT_NP;
            int st1, cn1, sd1, cls1, st2, cn2, sd2, cls2;
            param_rows(p1, st1, cn1, sd1, cls1);
            param_rows(p2, st2, cn2, sd2, cls2);
            double s = 0.0;
            for (int i = 0; i < cn1; ++i) {
                const int r1 = st1 + i * sd1;
                double inner = 0.0;
                for (int j = 0; j < cn2; ++j) { const int r2 = st2 + j * sd2; inner += sM[r1 * ZV + r2] * sJb[r2 * LIFT_NP + p2]; }
                s += sJa[r1 * LIFT_NP + p1] * inner;
            }
            hacc[slot] += s;
        }
        __syncthreads();
    }
    for (int k = tid, slot = 0; k < LIFT_NP * LIFT_NP; k += nthr, ++slot) out[(size_t)pair * LIFT_NP * LIFT_NP + k] = hacc[slot];
}

// ---------------------------------------------------------------------------------------------
// render_kernel: expected light of every source on one image, sum_s E_G_s.v in nanomaggies (the value-only
// add_pixel_term! sweep of bin/write_celeste_expectation.jl:112-156).  One wavefront per (source, chunk);
// contributions are added to the image plane with fp64 atomics (summation order, hence the last bits, is
// not deterministic).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
render_kernel(const DevPatch *__restrict__ patches, const double *__restrict__ coefs,
              const uint8_t *__restrict__ bitmaps, const float *__restrict__ pixels, const SrcImg *__restrict__ srcimg,
              const Comp *__restrict__ comps, int n, int N, int NC, int CH, int chunk_px, int imgH,
              double *__restrict__ plane, const int32_t *__restrict__ vis_off, const int32_t *__restrict__ vis_img,
              int dense) {
    __shared__ double etab[64];
    const int S = gridDim.x / CH;
    const int ch = blockIdx.x / S;
    const int s = blockIdx.x - ch * S;
    const int sn = table_entry(vis_off, vis_img, dense, N, s, n);
    if (sn < 0) return;                    // the source has no patch in this image
    const DevPatch &P = patches[sn];
    const int H2 = P.H2, W2 = P.W2;
    const int npx = H2 * (W2 - 1);
    const int p0 = ch * chunk_px;
    if (p0 >= npx) return;
    exp_table_init(etab);
    const int p1 = min(npx, p0 + chunk_px);
    const SrcImg si = srcimg[sn];
    __shared__ Comp tc[14 * CEL_MAXK];
    {
        const double *src = reinterpret_cast<const double *>(comps + (size_t)sn * NC);
        double *dst = reinterpret_cast<double *>(tc);
        for (int i = threadIdx.x; i < NC * 8; i += 64) dst[i] = src[i];
        __syncthreads();
    }
    const double *__restrict__ coef = coefs + (size_t)(CELESTE_MUTANT == 3 ? 0 : P.stamp) * (CEL_COEF * CEL_COEF);
    for (int idx = p0 + (int)threadIdx.x; idx < p1; idx += 64) {
        const int w2 = idx / H2, h2 = idx - w2 * H2;
        const size_t gi = (size_t)(P.off_h + h2) + (size_t)imgH * (P.off_w + w2);
        bool act = P.bitmap_off >= 0 ? bitmaps[P.bitmap_off + h2 + (int64_t)H2 * w2] != 0 : !isnan(pixels[gi]);
        if (!act) continue;
        const double hh = (double)(P.off_h + h2 + 1), ww = (double)(P.off_w + w2 + 1);
        const double f0 = star_value(coef, hh + (26.0 - si.m1), ww + (26.0 - si.m2));
        const double f1 = galaxy_value(tc, NC, hh - si.m1, ww - si.m2, etab);
        atomicAdd(&plane[gi], si.c0 * f0 + si.c1 * f1);
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
