// elbo_device.h -- device-side data layout shared by the kernels and the host ABI.
//
// HBM layout (all per-context, immutable after celeste_ctx_create unless noted):
//   DevImage[n_images]            image header + pointers to f32 planes (column-major, h fastest)
//   DevPatch[n_sources*n_images]  ImagePatch constants (imaged_sources.jl:60-71)
//   coefs[n_stamps][53*53]        conditioned + prefiltered cubic B-spline coefficients (f64)
//   bitmaps                       byte pool for explicit active_pixel_bitmaps (optional)
//   nbr_off / nbr_idx             neighbour CSR
// per batch (rewritten by prep_kernel every call, because vp changes every call):
//   SrcImg[n_sources*n_images]    pixel-space position + a_i E[l], a_i E[l^2] for the image's band
//   Comp[n_sources*n_images*NC]   PSF (x) galaxy-prototype bivariate normals, NC = 14*psf_K
//   SrcGeo[n_sources]             dXiXi/d(shape) Jacobian and second-derivative tensor
//   acc[n_targets*n_images*CH][ACC_N]  pixel-sum records in the reduced 10-variable space
#pragma once
#include <stdint.h>

#define CEL_P 44
#define CEL_MAXK 4
#define CEL_COEF 53
#define CEL_STAMP 51

// reduced variable order: c0 c1 q0 q1 | m1 m2 | dev | Xi11 Xi12 Xi22
#define ZV 10
// record: [0] value, [1..10] gradient, [11..65] packed upper triangle of the 10x10 Hessian,
// [66] active (pixel, source) pairs, [67] inactive pairs
#define ACC_N 68
#define ACC_H0 11
#define ACC_CNT 66

struct DevImage {
    int32_t H, W, band, pad;
    const float *pixels;
    const float *sky;
    const float *iota;
    // parameter-independent per-row term, computed once at context creation:
    const double *log_iota;  // Float32 log(iota[h]) widened (elbo_objective.jl:292)
};

struct DevPatch {
    int32_t off_h, off_w, H2, W2;
    int64_t bitmap_off;  // -1: bitmap == !isnan(pixel)
    int32_t stamp, pad;
    double J[4];         // wcs_jacobian, column-major
    double wc[2];        // world_center
    double pc[2];        // pixel_center
    double psf[CEL_MAXK * 6];
};

// one PSF component (x) one galaxy prototype component (fsm_util.jl:37-65)
struct Comp {
    double p11, p12, p22;  // precision = inv(tauBar_k + nuBar_j XiXi)
    double w0;             // z * gal_frac_dev_i      (f = w0 * exp(...))
    double wd;             // z * gal_frac_dev_dir    (d f / d gal_frac_dev = wd * exp(...))
    double nu;             // nuBar_j
    double xi1, xi2;       // xiBar_k: the mean is xi + m_pos (m_pos lives in SrcImg, so coordinates stay centred);
                           // last, because the pixel kernel reads it once per run of prototypes (3 x 16-byte reads per
                           // component instead of 4)
};                         // 64 bytes

struct SrcImg {
    double m1, m2;          // linear_world_to_pix(pos)
    double c0, c1, q0, q1;  // a_i E_l_a[b,i], a_i E_ll_a[b,i]
    double dev;             // gal_frac_dev (theta_0; theta_1 = 1 - dev)
    double pad1;
};

struct SrcGeo {
    double jsh[9];   // d(Xi11,Xi12,Xi22)/d(axis_ratio, angle, radius), column-major [sig + 3*shape]
    double tsh[27];  // second derivatives [sig + 3*s1 + 9*s2]
    int32_t finite;  // all 44 parameters finite
    int32_t pad;
};

__host__ __device__ constexpr int hidx(int i, int j) {  // i <= j, packed upper triangle of 10x10
    return ACC_H0 + i * ZV - (i * (i - 1)) / 2 + (j - i);
}
