// optim_kernels.h -- the caller of the ELBO hot path on the device.
//
//   reference                                                       here
//   --------------------------------------------------------------  -------------------------------------
//   ElboMaximize.elbo_constraints (ElboMaximize.jl:63-93)           box_bounds / simplex tables
//   ConstraintTransforms.enforce!, to_free!  (:84-126, :225-253)    optim_init_kernel
//   ConstraintTransforms.to_bound!           (:66-82, :128-160)     to_bound_dev
//   ConstraintTransforms.propagate_derivatives! (:373-457)          optim_step_kernel (analytic Jacobian/Hessian
//                                                                   of the transform instead of nested ForwardDiff)
//   ElboMaximize.maximize! -> Optim.optimize(NewtonTrustRegion)     optim_step_kernel + host loop in celeste_abi.hip
//       (ElboMaximize.jl:228-242; Optim.jl is third-party, unvendored: restated from its published algorithm,
//        Nocedal & Wright Alg. 4.1 / section 4.3)
//
// One 256-thread workgroup per target and Newton iteration: chain rule to the 41 free parameters, accept /
// reject + radius update, then the next trust-region sub-problem solved exactly in the eigenbasis of the
// 41 x 41 Hessian (parallel round-robin Jacobi rotations in LDS, secular equation by safeguarded Newton).
#pragma once
#include <hip/hip_runtime.h>
#include "elbo_device.h"
#include "../../include/celeste_mi355x.h"

#define NF 41
#define NJ 42  // Jacobi works on an even order; index 41 is an inert dummy

struct OptState {                 // per target slot
    double x[NF], xt[NF], g[NF];
    double f, delta, m;
    double pos0[2];               // centre of the position box (ElboMaximize.jl:70-73)
    int32_t iter, done, interior, evals, status, pad;
};

struct OptParams {
    double loc_width, loc_scale, xtol_abs, ftol_rel, gtol, initial_delta, delta_hat;
    int32_t max_iters, pad;
};

__device__ __forceinline__ void box_bounds(int i, const double *pos0, const OptParams &op, double &lo, double &hi,
                                           double &scale) {
    scale = 1.0;
    if (i < 2) { lo = pos0[i] - op.loc_width; hi = pos0[i] + op.loc_width; scale = op.loc_scale; }
    else if (i < 4) { lo = 1e-2; hi = 0.99; }
    else if (i == 4) { lo = -10.0; hi = 10.0; }
    else if (i == 5) { lo = 0.10; hi = 70.0; }
    else if (i < 8) { lo = -1.0; hi = 10.0; }
    else if (i < 10) { lo = 1e-4; hi = 0.10; }
    else if (i < 18) { lo = -10.0; hi = 10.0; }
    else { lo = 1e-4; hi = 1.0; }
}
__device__ __constant__ double c_simplex_lo[3] = {0.005, 0.01 / 8, 0.01 / 8};
__device__ __constant__ int c_simplex_n[3] = {2, 8, 8};
__device__ __constant__ int c_simplex_b0[3] = {26, 28, 36};
__device__ __constant__ int c_simplex_f0[3] = {26, 27, 34};

// to_bound! for one group of the simplex constraints: free x[f0..f0+n-2] -> p[0..n-1] (softmax with the last
// logit fixed at 0), bound = (1 - n lo) p + lo
__device__ inline void simplex_probs(const double *x, int g, double *p) {
    const int n = c_simplex_n[g], f0 = c_simplex_f0[g];
    double m = x[f0];
    for (int i = 1; i < n - 1; ++i) m = fmax(m, x[f0 + i]);
    const double exp_neg_m = exp(-m);
    double sum = exp_neg_m;
    for (int i = 0; i < n - 1; ++i) { p[i] = exp(x[f0 + i] - m); sum += p[i]; }
    for (int i = 0; i < n - 1; ++i) p[i] = p[i] / sum;
    p[n - 1] = (1.0 / sum) * exp_neg_m;
}

// serial to_bound (used by one thread per target): x (41) -> vs (44)
__device__ inline void to_bound_dev(const double *x, const double *pos0, const OptParams &op, double *vs) {
    for (int i = 0; i < 26; ++i) {
        double lo, hi, sc;
        box_bounds(i, pos0, op, lo, hi, sc);
        vs[i] = (1.0 / (1.0 + exp(-x[i] / sc))) * (hi - lo) + lo;
    }
    for (int g = 0; g < 3; ++g) {
        double p[8];
        simplex_probs(x, g, p);
        const int n = c_simplex_n[g];
        const double lo = c_simplex_lo[g];
        for (int i = 0; i < n; ++i) vs[c_simplex_b0[g] + i] = (1 - n * lo) * p[i] + lo;
    }
}

// enforce! + to_free! + first trial point; one thread per target
__global__ void optim_init_kernel(double *__restrict__ vp, const int32_t *__restrict__ targets, int n_targets,
                                  OptParams op, OptState *__restrict__ st, int32_t *__restrict__ active,
                                  const double *__restrict__ pos_centers) {
    const int ti = blockIdx.x * blockDim.x + threadIdx.x;
    if (ti >= n_targets) return;
    double *vs = vp + (size_t)targets[ti] * CEL_P;
    OptState &S = st[ti];
    // the position box stays where the first ElboConfig put it (ParallelRun.jl:96-100); default: current position
    S.pos0[0] = pos_centers ? pos_centers[2 * ti] : vs[0];
    S.pos0[1] = pos_centers ? pos_centers[2 * ti + 1] : vs[1];
    for (int i = 0; i < 26; ++i) {
        double lo, hi, sc;
        box_bounds(i, S.pos0, op, lo, hi, sc);
        double b = vs[i];
        if (!(lo < b && b < hi)) b = fmax(fmin(b, nextafter(hi, -INFINITY)), nextafter(lo, INFINITY));
        vs[i] = b;
        S.x[i] = -log(1.0 / ((b - lo) / (hi - lo)) - 1.0) * sc;
    }
    for (int g = 0; g < 3; ++g) {
        const int n = c_simplex_n[g], b0 = c_simplex_b0[g], f0 = c_simplex_f0[g];
        const double lo = c_simplex_lo[g];
        double sum = 0;
        for (int i = 0; i < n; ++i) {
            double b = vs[b0 + i];
            if (!(lo < b && b < 1.0)) b = fmax(fmin(b, nextafter(1.0, -INFINITY)), nextafter(lo, INFINITY));
            vs[b0 + i] = b; sum += b;
        }
        if (!(fabs(sum - 1.0) <= 1.4901161193847656e-08 * fmax(fabs(sum), 1.0))) {
            const double rescale = (1 - n * lo) / (sum - n * lo);
            for (int i = 0; i < n; ++i) vs[b0 + i] = nextafter(lo, INFINITY) + rescale * (vs[b0 + i] - lo);
        }
        const double log_last = log((vs[b0 + n - 1] - lo) / (1 - n * lo));
        for (int i = 0; i < n - 1; ++i) S.x[f0 + i] = log((vs[b0 + i] - lo) / (1 - n * lo)) - log_last;
    }
    for (int i = 0; i < NF; ++i) S.xt[i] = S.x[i];
    S.f = 0; S.delta = op.initial_delta; S.m = 0; S.iter = -1; S.done = 0; S.interior = 0; S.evals = 0; S.status = 0;
    to_bound_dev(S.xt, S.pos0, op, vs);  // the evaluation point, as evaluate! does (ElboMaximize.jl:163-166)
    active[ti] = ti;
}

// ---------------------------------------------------------------------------------------------------------
// optim_step_kernel
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
optim_step_kernel(double *__restrict__ vp, const int32_t *__restrict__ targets, const int32_t *__restrict__ active,
                  const double *__restrict__ ev_v, const double *__restrict__ ev_d, const double *__restrict__ ev_h,
                  const int32_t *__restrict__ ev_status, OptParams op, OptState *__restrict__ st,
                  double *__restrict__ Hstate, int32_t *__restrict__ next_active, int32_t *__restrict__ next_targets,
                  int32_t *__restrict__ next_count) {
    __shared__ double sA[NJ * NJ];        // Hessian being diagonalised (column-major)
    __shared__ double sV[NJ * NJ];        // eigenvectors
    __shared__ double sHt[NF * NF];       // trial-point Hessian (negated, free space)
    __shared__ double sd[CEL_P], sx[NF], sg[NF], sgt[NF], sJb[26], sHb[26], sp[3][8];
    __shared__ double sc_[21], ss_[21];   // rotation cos / sin of the current round
    __shared__ int spq[21][2];
    __shared__ double sw[NJ], sqg[NJ], scv[NJ], sred[8];
    __shared__ int s_flag[4];             // 0: accept, 1: done, 2: jacobi converged

    const int li = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    const int slot = active[li];
    OptState &S = st[slot];
    const int t = targets[slot];
    double *Hs = Hstate + (size_t)slot * NF * NF;
    const double *h = ev_h + (size_t)li * CEL_P * CEL_P;

    // ---- chain rule to the free parameters at the evaluated point xt (propagate_derivatives!) ----
    if (tid < CEL_P) sd[tid] = ev_d[(size_t)li * CEL_P + tid];
    if (tid < NF) sx[tid] = S.xt[tid];
    __syncthreads();
    if (tid < 26) {
        double lo, hi, sc;
        box_bounds(tid, S.pos0, op, lo, hi, sc);
        const double s = 1.0 / (1.0 + exp(-sx[tid] / sc)), w = hi - lo;
        sJb[tid] = w * s * (1 - s) / sc;
        sHb[tid] = w * s * (1 - s) * (1 - 2 * s) / (sc * sc);
    } else if (tid < 29) simplex_probs(sx, tid - 26, sp[tid - 26]);
    __syncthreads();
    // Jacobian entry d bound_a / d free_i (0 outside the parameter's own constraint group)
    auto jac = [&](int a, int i) -> double {
        if (i < 26) return a == i ? sJb[i] : 0.0;
        const int g = i < 27 ? 0 : (i < 34 ? 1 : 2);
        const int b0 = c_simplex_b0[g], n = c_simplex_n[g], j = i - c_simplex_f0[g];
        if (a < b0 || a >= b0 + n) return 0.0;
        const double *p = sp[g];
        return (1 - n * c_simplex_lo[g]) * p[a - b0] * (((a - b0) == j) - p[j]);
    };
    auto group_lo = [&](int i) { return i < 26 ? i : c_simplex_b0[i < 27 ? 0 : (i < 34 ? 1 : 2)]; };
    auto group_n = [&](int i) { return i < 26 ? 1 : c_simplex_n[i < 27 ? 0 : (i < 34 ? 1 : 2)]; };
    if (tid < NF) {
        double s = 0;
        const int a0 = group_lo(tid), na = group_n(tid);
        for (int a = a0; a < a0 + na; ++a) s += jac(a, tid) * sd[a];
        sgt[tid] = -s;  // minimise -elbo
    }
    for (int k = tid; k < NF * NF; k += nthr) {
        const int j = k / NF, i = k - j * NF;
        if (i > j) continue;
        const int a0 = group_lo(i), na = group_n(i), b0 = group_lo(j), nb = group_n(j);
        double s = 0;
        for (int a = a0; a < a0 + na; ++a) {
            const double ja = jac(a, i);
            double inner = 0;
            for (int b = b0; b < b0 + nb; ++b) inner += h[a + CEL_P * b] * jac(b, j);
            s += ja * inner;
        }
        // second derivatives of the transform, contracted with the bound gradient
        if (i < 26) { if (i == j) s += sd[i] * sHb[i]; }
        else if (a0 == b0) {
            const int g = i < 27 ? 0 : (i < 34 ? 1 : 2);
            const int n = c_simplex_n[g], f0 = c_simplex_f0[g], jj = i - f0, kk = j - f0;
            const double *p = sp[g];
            const double scl = 1 - n * c_simplex_lo[g];
            for (int a = 0; a < n; ++a) {
                const double d2 = p[a] * (((a == jj) - p[jj]) * ((a == kk) - p[kk]) - p[jj] * ((jj == kk) - p[kk]));
                s += sd[a0 + a] * scl * d2;
            }
        }
        sHt[i + NF * j] = -s; sHt[j + NF * i] = -s;
    }
    __syncthreads();

    // ---- accept / reject, radius update, convergence (N&W Alg. 4.1 as in Optim.jl's NewtonTrustRegion) ----
    if (tid == 0) {
        const double ft = -ev_v[li];
        int accept = 1, done = 0;
        S.evals += 1;
        if (ev_status[li] != CELESTE_OK) { S.status = ev_status[li]; accept = 0; done = 1; }
        else if (S.iter >= 0) {
            const double m = S.m;
            double rho;
            if (fabs(m) <= 2.220446049250313e-16) rho = 1.0;
            else if (m > 0) rho = 0.25 - 1.0;
            else rho = (S.f - ft) / (0 - m);
            if (rho < 0.25) S.delta *= 0.25;
            else if (rho > 0.75 && !S.interior) S.delta = fmin(2 * S.delta, op.delta_hat);
            accept = rho > 0.1;
            if (accept) {
                double dx = 0, gmax = 0;
                for (int i = 0; i < NF; ++i) { dx = fmax(dx, fabs(sx[i] - S.x[i])); gmax = fmax(gmax, fabs(sgt[i])); }
                if (dx <= op.xtol_abs || fabs(ft - S.f) <= op.ftol_rel * fabs(ft) || gmax <= op.gtol) done = 1;
            }
        }
        if (accept) S.f = ft;
        S.iter += 1;
        if (S.iter >= op.max_iters) done = 1;
        s_flag[0] = accept; s_flag[1] = done;
    }
    __syncthreads();
    const int accept = s_flag[0], done = s_flag[1];
    if (accept) {
        if (tid < NF) { S.x[tid] = sx[tid]; S.g[tid] = sgt[tid]; sg[tid] = sgt[tid]; }
        for (int k = tid; k < NF * NF; k += nthr) Hs[k] = sHt[k];
    } else {
        if (tid < NF) { sx[tid] = S.x[tid]; sg[tid] = S.g[tid]; }
        for (int k = tid; k < NF * NF; k += nthr) sHt[k] = Hs[k];
    }
    __syncthreads();
    if (done) {
        if (tid == 0) { S.done = 1; to_bound_dev(S.x, S.pos0, op, vp + (size_t)t * CEL_P); }
        return;
    }

    // ---- trust-region sub-problem at the accepted point: eigen-decomposition by parallel Jacobi ----
    for (int k = tid; k < NJ * NJ; k += nthr) {
        const int j = k / NJ, i = k - j * NJ;
        sA[k] = (i < NF && j < NF) ? sHt[i + NF * j] : 0.0;
        sV[k] = (i == j) ? 1.0 : 0.0;
    }
    __syncthreads();
    for (int sweep = 0; sweep < 30; ++sweep) {
        // convergence: off-diagonal mass against the diagonal
        double off = 0, dg = 0;
        for (int k = tid; k < NF * NF; k += nthr) {
            const int j = k / NF, i = k - j * NF;
            const double a = sA[i + NJ * j];
            if (i == j) dg += a * a; else off += a * a;
        }
        for (int o = 32; o >= 1; o >>= 1) { off += __shfl_xor(off, o, 64); dg += __shfl_xor(dg, o, 64); }
        if ((tid & 63) == 0) { sred[tid >> 6] = off; sred[4 + (tid >> 6)] = dg; }
        __syncthreads();
        if (tid == 0) {
            const double o = sred[0] + sred[1] + sred[2] + sred[3], d = sred[4] + sred[5] + sred[6] + sred[7];
            s_flag[2] = (o <= 1e-30 * (o + d)) || o == 0.0;
        }
        __syncthreads();
        if (s_flag[2]) break;
        for (int r = 0; r < NJ - 1; ++r) {
            // round-robin pairing: player 41 (the dummy) is fixed; its pair is skipped
            if (tid < 20) {
                const int k = tid + 1;
                int p = (r + k) % (NJ - 1), q = (r - k + (NJ - 1)) % (NJ - 1);
                if (p > q) { const int u = p; p = q; q = u; }
                const double apq = sA[p + NJ * q];
                double c = 1.0, s = 0.0;
                if (apq != 0.0) {
                    const double theta = (sA[q + NJ * q] - sA[p + NJ * p]) / (2 * apq);
                    const double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                    c = 1 / sqrt(tt * tt + 1); s = tt * c;
                }
                sc_[tid] = c; ss_[tid] = s; spq[tid][0] = p; spq[tid][1] = q;
            }
            __syncthreads();
            for (int k = tid; k < 20 * NF; k += nthr) {  // rows p, q
                const int pr = k / NF, col = k - pr * NF;
                const int p = spq[pr][0], q = spq[pr][1];
                const double c = sc_[pr], s = ss_[pr];
                const double ap = sA[p + NJ * col], aq = sA[q + NJ * col];
                sA[p + NJ * col] = c * ap - s * aq; sA[q + NJ * col] = s * ap + c * aq;
            }
            __syncthreads();
            for (int k = tid; k < 20 * NF; k += nthr) {  // columns p, q of A and of V
                const int pr = k / NF, row = k - pr * NF;
                const int p = spq[pr][0], q = spq[pr][1];
                const double c = sc_[pr], s = ss_[pr];
                const double ap = sA[row + NJ * p], aq = sA[row + NJ * q];
                sA[row + NJ * p] = c * ap - s * aq; sA[row + NJ * q] = s * ap + c * aq;
                const double vp_ = sV[row + NJ * p], vq = sV[row + NJ * q];
                sV[row + NJ * p] = c * vp_ - s * vq; sV[row + NJ * q] = s * vp_ + c * vq;
            }
            __syncthreads();
        }
    }
    if (tid < NF) {
        sw[tid] = sA[tid + NJ * tid];
        double q = 0;
        for (int k = 0; k < NF; ++k) q += sV[k + NJ * tid] * sg[k];
        sqg[tid] = q;
    }
    __syncthreads();
    if (tid == 0) {
        // exact sub-problem solution in the eigenbasis (N&W section 4.3), hard case included
        double wmin = sw[0], wmax = sw[0];
        int imin = 0;
        for (int i = 1; i < NF; ++i) { if (sw[i] < wmin) { wmin = sw[i]; imin = i; } wmax = fmax(wmax, sw[i]); }
        const double delta = S.delta, d2 = delta * delta;
        int interior = 0;
        if (wmin >= 1e-8) {
            double p2 = 0;
            for (int i = 0; i < NF; ++i) p2 += (sqg[i] / sw[i]) * (sqg[i] / sw[i]);
            interior = p2 <= d2;
        }
        if (interior) {
            for (int i = 0; i < NF; ++i) scv[i] = -sqg[i] / sw[i];
        } else {
            const double lambda_lb = -wmin + fmax(1e-8, 1e-8 * (wmax - wmin));
            double lambda = fmax(lambda_lb, 0.0), p2 = 0;
            for (int i = 0; i < NF; ++i) { const double r = sqg[i] / (sw[i] + lambda); p2 += r * r; }
            if (p2 < d2) {
                for (int i = 0; i < NF; ++i) scv[i] = -sqg[i] / (sw[i] + lambda);
                const double tau = sqrt(d2 - p2);
                scv[imin] += (scv[imin] >= 0 ? tau : -tau);
            } else {
                for (int it = 0; it < 100; ++it) {
                    double q2 = 0, q3 = 0;
                    for (int i = 0; i < NF; ++i) { const double r = sqg[i] / (sw[i] + lambda); q2 += r * r; q3 += r * r / (sw[i] + lambda); }
                    const double nrm = sqrt(q2);
                    double ln = lambda + (q2 / q3) * (nrm - delta) / delta;
                    if (ln < lambda_lb) ln = 0.5 * (lambda + lambda_lb);
                    const bool conv = fabs(ln - lambda) <= 1e-12 * fmax(1.0, fabs(ln));
                    lambda = ln;
                    if (conv) break;
                }
                for (int i = 0; i < NF; ++i) scv[i] = -sqg[i] / (sw[i] + lambda);
            }
        }
        double m = 0;
        for (int i = 0; i < NF; ++i) m += sqg[i] * scv[i] + 0.5 * sw[i] * scv[i] * scv[i];
        S.m = m; S.interior = interior;
    }
    __syncthreads();
    if (tid < NF) {
        double s = 0;
        for (int i = 0; i < NF; ++i) s += sV[tid + NJ * i] * scv[i];
        const double xn = sx[tid] + s;
        S.xt[tid] = xn; sx[tid] = xn;
    }
    __syncthreads();
    if (tid == 0) {
        to_bound_dev(sx, S.pos0, op, vp + (size_t)t * CEL_P);   // next evaluation point
        const int pos = atomicAdd(next_count, 1);
        next_active[pos] = slot; next_targets[pos] = t;
    }
}
