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
// One wavefront (64-thread workgroup) per target and Newton iteration: chain rule to the 41 free parameters,
// accept / reject + radius update, then the next trust-region sub-problem (N&W section 4.3).  The exact
// solution p(lambda) = -(H + lambda I)^-1 g only needs H up to an orthogonal change of basis, so H is reduced to
// tridiagonal form T = Q' H Q by Householder reflections (lanes = rows, 13 KB of LDS) and everything else runs
// in O(n) per lambda: extreme eigenvalues of T by 64-way Sturm multisection, (T + lambda I) y = -Q'g by LDL',
// the secular equation by safeguarded Newton, p = Q y.  Same step as the eigen-decomposition Optim.jl uses, to
// the tolerance of the secular solve, at a fraction of its latency (the hard case included: lowest eigenvectors
// by inverse iteration); OptParams.solver = 1 takes the full eigen-decomposition (implicit-shift QL) instead.
// Sub-problem rules (Optim.jl's solve_tr_subproblem!, restated; Optim is third-party and not vendored): the plain
// Newton step when the smallest eigenvalue is >= 1e-8 and the step fits; otherwise lambda starts at
// lambda_lb = -w_min + max(1e-8, 1e-8 (w_max - w_min)); the hard case only when w_min < 0 and g is orthogonal
// (1e-10) to every eigenvector whose eigenvalue lies within 1e-10 of w_min; else Newton on lambda (tolerance 1e-10,
// halving towards lambda_lb on undershoot) run to convergence, the step being the last factorisation's.
#pragma once
#include <hip/hip_runtime.h>
#include "elbo_device.h"
#include "../../include/celeste_mi355x.h"

#include <type_traits>
#define NF 41
#ifndef OPTIM_UNROLL_RECURRENCES
#define OPTIM_UNROLL_RECURRENCES 1
#endif
#if OPTIM_UNROLL_RECURRENCES
#define OPTIM_UNROLL _Pragma("unroll")
#else
#define OPTIM_UNROLL
#endif
#ifndef OPTIM_TRED_LDS
#define OPTIM_TRED_LDS 0   // 1: the LDS-resident tridiagonalisation (tred_wave) in the trust-region solve too
#endif
#ifndef OPTIM_DPP_BOUND_CTRL
#define OPTIM_DPP_BOUND_CTRL 1
#endif
#ifndef OPTIM_TRED_SPLIT
#define OPTIM_TRED_SPLIT 1    // workgroups of four wavefronts: Q'g by wavefront 1 while wavefront 0 finds the extreme eigenvalues
#endif
#ifndef OPTIM_TRED_SQRT
#define OPTIM_TRED_SQRT 1     // the reduction's square roots without the library's range scaling
#endif
#ifndef OPTIM_HARD_SCREEN
#define OPTIM_HARD_SCREEN 1   // 0: run the eigenvector test of the hard case whenever the smallest eigenvalue is negative
#endif
#ifndef OPTIM_Q3_SOLVE
#define OPTIM_Q3_SOLVE 0      // 1: y' (T + lambda I)^-1 y by a second full solve instead of the forward sweep
#endif
#define SPEC_STATE (NF + 3)                // trial point, model decrease, interior flag, iteration tag; two of them per target,
                                           // used in turn (a slow writer can never meet the next one in the same block)
#define TRI_STATE (NF * NF + 4 * NF + 4)   // reflectors, (td, te, hv, Q'g) per row, wmin, wmax, wmin_lower, norm_bound
#define LDA 45  // leading dimension of the LDS matrix (holds the 44 x 44 bound-space Hessian first; odd: no bank conflicts)

struct OptState {                 // per target slot
    double x[NF], xt[NF], g[NF];
    double f, delta, m;
    double pos0[2];               // centre of the position box (ElboMaximize.jl:70-73)
    int32_t iter, done, interior, evals, status, pad;
};

struct OptParams {
    double loc_width, loc_scale, xtol_abs, ftol_rel, gtol, initial_delta, delta_hat;
    int32_t max_iters, solver;    // 0: tridiagonal-space solve (default), 1: eigen-decomposition always
    int32_t secular_iters, pad;   // cap of the Newton iterations on lambda (20: to convergence; Optim.jl stops after 5)
};

template <class OP>   // OptParams in any address space (the fused kernel reads it in the kernel-argument segment)
__device__ __forceinline__ void box_bounds(int i, const double *pos0, const OP &op, double &lo, double &hi,
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
// The same by eight lanes per group (lane = entry i of group g; the three groups side by side): the exponentials -- the
// expensive part, eight library calls in a row on one lane above -- in parallel, the sum by one lane in simplex_probs's order,
// the divisions in parallel: the same operations on the same operands, so the same bits.  x: LDS; ebuf[3][8], sbuf[3]: LDS
// scratch of the calling wavefront.  Returns p[i] of group g (lanes i >= n: nothing meaningful).
__device__ __forceinline__ double simplex_prob_lane(const double *x, int g, int i, double *ebuf, double *sbuf) {
    const int n = c_simplex_n[g], f0 = c_simplex_f0[g];
    double m = x[f0];
    for (int k = 1; k < n - 1; ++k) m = fmax(m, x[f0 + k]);
    const double e = i < n - 1 ? exp(x[f0 + i] - m) : exp(-m);    // (i = n - 1: the last entry's exp(-m))
    if (i < n) ebuf[8 * g + i] = e;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (i == 0) {
        double sum = ebuf[8 * g + n - 1];
        for (int k = 0; k < n - 1; ++k) sum += ebuf[8 * g + k];
        sbuf[g] = sum;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const double sum = sbuf[g];
    return i < n - 1 ? e / sum : (1.0 / sum) * e;
}

template <class OP>
__device__ inline void to_bound_dev(const double *x, const double *pos0, const OP &op, double *vs) {
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
// One thread: the start of maximize! for one target -- vs: its 44 parameters (enforced in place, then set to the first
// evaluation point), S: its optimiser state, pos: the centre of its position box (nullptr: its current position)
template <class OP>
__device__ inline void optim_init_values(double *__restrict__ vs, const OP &op, const double *__restrict__ pos, OptState &S) {
    // the position box stays where the first ElboConfig put it (ParallelRun.jl:96-100); default: current position
    S.pos0[0] = pos ? pos[0] : vs[0];
    S.pos0[1] = pos ? pos[1] : vs[1];
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
}

__global__ void optim_init_kernel(double *__restrict__ vp, const int32_t *__restrict__ targets, int n_targets,
                                  OptParams op, OptState *__restrict__ st, int32_t *__restrict__ active,
                                  const double *__restrict__ pos_centers) {
    const int ti = blockIdx.x * blockDim.x + threadIdx.x;
    if (ti >= n_targets) return;
    optim_init_values(vp + (size_t)targets[ti] * CEL_P, op, pos_centers ? pos_centers + 2 * ti : nullptr, st[ti]);
    active[ti] = ti;
}

// The trust-region solvers below are written for ONE wavefront (lane = row / element): lanes exchange data through LDS
// and registers of that wavefront only.  Its LDS instructions execute in order, so a hand-over between lanes needs no
// hardware barrier -- only that the compiler keeps the accesses in program order and does not carry LDS values in
// registers across the point.  (The callers' workgroups may have more wavefronts -- optim_fused_kernel's have four --
// which do not take part in the solve: a workgroup barrier in here would wait for them.)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ __forceinline__ double lane_bcast(double x, int src) { return __shfl(x, src, 64); }
// the same for a wave-uniform source lane: v_readlane into a scalar register pair instead of an LDS permute
__device__ __forceinline__ double lane_bcast_u(double x, int src) {
    const long long b = __builtin_bit_cast(long long, x);
    const int lo = __builtin_amdgcn_readlane((int)b, src), hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

// wave sum by DPP row operations (no LDS traffic): quad butterflies, row mirrors, row broadcasts; the total
// ends in lane 63 and is handed to every lane through a scalar register pair
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_add_f64(double x) {
    const long long b = __builtin_bit_cast(long long, x);
    // all rows written and every lane has a source (quad permutations, mirrors): bound_ctrl, so that the compiler need not
    // clear the destination first (two v_mov per level; the row broadcasts below do need their zeros)
    constexpr bool BC = ROWMASK == 0xF && OPTIM_DPP_BOUND_CTRL;
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, ROWMASK, 0xF, BC);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROWMASK, 0xF, BC);
    return x + __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
// Sum over the wavefront of a value that is ZERO in the lanes >= N (N static), the same in every lane afterwards.  Inside a
// row of 16 lanes: quad butterflies and row mirrors (DPP, no LDS traffic) -- levels that would only add zeros are left out;
// the row sums then travel through scalar registers and are combined as (r3 + r2) + (r1 + r0).  Every operation of the
// sum over all 64 lanes that is left out here adds an exact zero, so the result does not depend on N.  (The chains of
// dependent FP64 instructions are what the sub-problem's time is made of -- ~25 cycles each, measured on the Sturm
// recurrences: a sum over <= 16 lanes is 4 dependent additions instead of 6.)
template <int N>
__device__ __forceinline__ double wave_sum_n(double x) {
    if constexpr (N > 1) x = dpp_add_f64<0xB1, 0xF>(x);    // quad_perm [1,0,3,2]
    if constexpr (N > 2) x = dpp_add_f64<0x4E, 0xF>(x);    // quad_perm [2,3,0,1]
    if constexpr (N > 4) x = dpp_add_f64<0x141, 0xF>(x);   // row_half_mirror
    if constexpr (N > 8) x = dpp_add_f64<0x140, 0xF>(x);   // row_mirror: every lane holds its 16-lane row sum
    const double r0 = lane_bcast_u(x, 0);
    if constexpr (N <= 16) return r0;
    const double r1 = lane_bcast_u(x, 16);
    if constexpr (N <= 32) return r1 + r0;
    const double r2 = lane_bcast_u(x, 32);
    if constexpr (N <= 48) return r2 + (r1 + r0);
    const double r3 = lane_bcast_u(x, 48);
    return (r3 + r2) + (r1 + r0);
}
__device__ __forceinline__ double wave_sum_dpp(double x) { return wave_sum_n<64>(x); }

// Householder reduction of the symmetric NF x NF matrix A (LDS, column-major, full storage) to tridiagonal
// form.  On exit: row i of A (columns < i) holds the Householder vector u_i, hv[i] = |u_i|^2 / 2 (0: no
// reflection), e[i] the sub-diagonal entry (i - 1, i), A(i, i) the diagonal.  q: NF doubles of scratch.
__device__ inline void tred_wave(double *A, double *hv, double *e, double *q, int ln) {
    for (int i = NF - 1; i >= 1; --i) {
        const int l = i - 1;
        const bool act = ln <= l;
        const double x = act ? A[i + LDA * ln] : 0.0;     // row i left of the diagonal
        if (l == 0) { e[i] = lane_bcast_u(x, 0); hv[i] = 0.0; continue; }
        // (no rescaling of the row: the entries are curvatures of an O(1e7) objective in O(1) free parameters, far
        // from the overflow / underflow range the classical algorithm guards against)
        const double f = lane_bcast_u(x, l);
        const double hoff = wave_sum_dpp(ln < l ? x * x : 0.0);
        if (hoff == 0.0) { e[i] = f; hv[i] = 0.0; continue; }   // nothing left of column l: already tridiagonal here
        double h = hoff + f * f;
        const double g = f >= 0 ? -sqrt(h) : sqrt(h);
        h -= f * g;
        const double u = (ln == l) ? f - g : x;           // Householder vector (0 beyond l)
        if (act) A[i + LDA * ln] = u;
        e[i] = g; hv[i] = h;
        wave_sync();
        double acc0 = 0.0, acc1 = 0.0;                    // two chains: the FMA latency is the critical path here
        if (act) {
            int k = 0;
            for (; k + 1 <= l; k += 2) {
                acc0 = __builtin_fma(A[ln + LDA * k], A[i + LDA * k], acc0);
                acc1 = __builtin_fma(A[ln + LDA * (k + 1)], A[i + LDA * (k + 1)], acc1);
            }
            if (k <= l) acc0 = __builtin_fma(A[ln + LDA * k], A[i + LDA * k], acc0);
        }
        const double rh = 1.0 / h;
        const double p = (acc0 + acc1) * rh;
        const double hh = wave_sum_dpp(p * u) * (0.5 * rh);
        const double qv = p - hh * u;
        if (act) q[ln] = qv;
        wave_sync();
        if (act) for (int k = 0; k <= l; ++k) A[ln + LDA * k] -= u * q[k] + qv * A[i + LDA * k];
        wave_sync();
    }
    hv[0] = 0.0; e[0] = 0.0;
    wave_sync();
}

// ---------------------------------------------------------------------------------------------------------
// The same reduction with the matrix in registers: lane ln holds row ln (every index static: the NF - 1 steps are unrolled
// by template recursion).  The LDS version could not hide its 4 (l + 1) LDS round trips per step (76 us per matrix alone on
// a CU; measured with -DOPTIM_TIMING).
//
// How entry k of the Householder vector u (and of q) reaches the FMAs of every lane decides the cost: a wavefront alone on
// its SIMD -- the sub-problem of a fused launch -- issues one FP64 instruction per ~6 cycles whatever their dependencies
// (tools/fp64_rate_probe.hip 1), so the reduction's time is its instruction count.  Through scalar registers (two
// v_readlane per use) a column costs six of them per step next to its three FMAs: 12 k instructions, 58.4 k cycles
// (tools/tred_probe.hip).  Here the vector is laid out once per step so that every row of 16 lanes holds it
// (row_replicate) and the FMAs take their operand through DPP row_newbcast: three instructions per column, 50.7 k cycles
// with the same bits; without the third per-step vector, the library square root's range scaling and (in the fused launch)
// the reflections of g: see the figures in DESIGN.md section 6c.
// (Tried and dropped, round 4: the same reduction shared out over the four wavefronts of a fused workgroup -- columns
// k = w mod 4 per wavefront, one LDS exchange and one barrier per step, bit-identical.  56.9 k cycles against 58.4 k: the
// exchange costs 410 of a step's 1450 cycles, as much as the column work it takes off each wavefront.)
// Out: row i of the LDS matrix A (columns < i) = u_i (for p = Q y later); in lane ln: td = diagonal entry ln,
// ev = sub-diagonal entry (ln - 1, ln), hvv = |u_ln|^2 / 2 (0: no reflection).
// ---------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) double lds_double_t;
__device__ __forceinline__ lds_double_t *as_lds(double *p) { return (lds_double_t *)p; }

// ---- lane broadcasts without v_readlane: row-replicated copies + DPP row_newbcast ----
// On gfx90a and later v_fmac_f64 takes its first operand through DPP, and row_newbcast:n hands lane n of each row of 16 to
// the whole row.  For that the vector must stand in every row: R.r[j] holds, in all four rows, row j of the vector (lanes
// 16 j ... 16 j + 15) -- three swaps per dword (v_permlane16_swap puts rows (0,0,2,2) and (1,1,3,3) side by side,
// v_permlane32_swap completes them).  Same products, same sums as with a scalar operand: the results do not change.
// The DPP read needs two wait states behind a VALU write of its operand, and the compiler does not see a DPP instruction
// in the asm: the s_nop below covers the swaps, and tests/test_dpp_hazard.py checks the compiled code for any other
// write the register allocator may have put in between.
struct RowRep { double r[3]; };
// ROWS: the rows of 16 lanes that will look at the copies, 4 = all of them (the tridiagonal recurrences: every lane carries the
// whole chain).  The reduction's lanes beyond row l / 16 only hold zeros of u and q and finished parts of the matrix, so what
// they multiply does not matter as long as it is finite: with ROWS = 1 the vector itself serves (row 0 reads its own lanes),
// with ROWS = 2 one v_permlane16_swap per dword does (rows 0 and 1 then hold row 0 in one register and row 1 in the other).
template <int ROWS>
__device__ __forceinline__ void row_rep_dword(unsigned d, unsigned &o0, unsigned &o1, unsigned &o2) {
    if constexpr (ROWS == 1) { o0 = d; o1 = 0u; o2 = 0u; }
    else {
        const auto p = __builtin_amdgcn_permlane16_swap(d, d, false, false);        // p[0] = rows (0,0,2,2), p[1] = rows (1,1,3,3)
        if constexpr (ROWS == 2) { o0 = p[0]; o1 = p[1]; o2 = 0u; }
        else {
            const auto q = __builtin_amdgcn_permlane32_swap(p[0], p[0], false, false);  // q[0] = rows (0,0,0,0), q[1] = rows (2,2,2,2)
            o0 = q[0]; o2 = q[1];
            o1 = __builtin_amdgcn_permlane32_swap(p[1], p[1], false, false)[0];     // rows (1,1,1,1)
        }
    }
}
template <int ROWS>
__device__ __forceinline__ RowRep row_replicate(double x) {
    const long long b = __builtin_bit_cast(long long, x);
    unsigned l0, l1, l2, h0, h1, h2;
    row_rep_dword<ROWS>((unsigned)b, l0, l1, l2);
    row_rep_dword<ROWS>((unsigned)(b >> 32), h0, h1, h2);
    RowRep R;
    R.r[0] = __builtin_bit_cast(double, ((unsigned long long)h0 << 32) | l0);
    R.r[1] = __builtin_bit_cast(double, ((unsigned long long)h1 << 32) | l1);
    R.r[2] = __builtin_bit_cast(double, ((unsigned long long)h2 << 32) | l2);
    if constexpr (ROWS >= 3) asm volatile("s_nop 1" : "+v"(R.r[0]), "+v"(R.r[1]), "+v"(R.r[2]));
    else if constexpr (ROWS == 2) asm volatile("s_nop 1" : "+v"(R.r[0]), "+v"(R.r[1]));
    else asm volatile("s_nop 1" : "+v"(R.r[0]));
    return R;
}
// acc += vec[K] * other, vec given by its row-replicated copies
template <int K>
__device__ __forceinline__ void fmac_bcast(double &acc, const RowRep &R, double other) {
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(R.r[K / 16]), "v"(other), "n"(K % 16));
}
// vec[K] in every lane (one v_mov_b64 through DPP instead of two v_readlane and a scalar operand)
template <int K>
__device__ __forceinline__ double bcast_mov(const RowRep &R) {
    double o;
    asm("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(R.r[K / 16]), "n"(K % 16));
    return o;
}
// f(integral_constant<int, k>) for k = K ... N - 1 (static_for) / k = K ... END downwards (static_for_down): unrolled loops
// whose index the asm above can take as an immediate
template <int K, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (K < N) { f(std::integral_constant<int, K>{}); static_for<K + 1, N>(f); }
}
template <int K, int END, class F>
__device__ __forceinline__ void static_for_down(F &&f) {
    if constexpr (K >= END) { f(std::integral_constant<int, K>{}); static_for_down<K - 1, END>(f); }
}
template <int K, int L>   // A u, columns K ... L, four chains by k mod 4
__device__ __forceinline__ void tred_matvec_bc(const double (&a)[NF], const RowRep &U, double &c0, double &c1, double &c2, double &c3) {
    if constexpr (K <= L) {
        if constexpr ((K & 3) == 0) fmac_bcast<K>(c0, U, a[K]);
        else if constexpr ((K & 3) == 1) fmac_bcast<K>(c1, U, a[K]);
        else if constexpr ((K & 3) == 2) fmac_bcast<K>(c2, U, a[K]);
        else fmac_bcast<K>(c3, U, a[K]);
        tred_matvec_bc<K + 1, L>(a, U, c0, c1, c2, c3);
    }
}
template <int K>          // a_k <- (a_k - u q_k) - q u_k for the columns K ... 0 (nu, nq: -u and -q of the lane)
__device__ __forceinline__ void tred_update_bc(double (&a)[NF], const RowRep &U, const RowRep &Q, double nu, double nq) {
    if constexpr (K >= 0) {
        fmac_bcast<K>(a[K], Q, nu);
        fmac_bcast<K>(a[K], U, nq);
        tred_update_bc<K - 1>(a, U, Q, nu, nq);
    }
}

// 1 / h (h > 0): hardware reciprocal + two Newton steps -- half the dependent instructions of an IEEE division
__device__ __forceinline__ double tred_rcp(double h) {
    double rh = __builtin_amdgcn_rcp(h);
    rh = __builtin_fma(__builtin_fma(-h, rh, 1.0), rh, rh);
    return __builtin_fma(__builtin_fma(-h, rh, 1.0), rh, rh);
}
// sqrt(h), h uniform over the wavefront: the library's iteration (v_rsq_f64, one coupled Newton step, two corrections)
// without its range scaling and special-value selects while h is far from the ends of the exponent range -- always, for
// sums of squares of Hessian entries, unless a parameter sits so deep in the flat part of its transform that its row
// underflows; then the library's.
__device__ __forceinline__ double tred_sqrt(double h) {
#if OPTIM_TRED_SQRT
    if (h > 1e-280 && h < 1e280) {
        const double y = __builtin_amdgcn_rsq(h);
        double g = h * y, hy = 0.5 * y;
        const double r = __builtin_fma(-hy, g, 0.5);
        g = __builtin_fma(g, r, g); hy = __builtin_fma(hy, r, hy);
        g = __builtin_fma(__builtin_fma(-g, g, h), hy, g);
        return __builtin_fma(__builtin_fma(-g, g, h), hy, g);
    }
#endif
    return sqrt(h);
}
// v <- (I - u u' / h) v, one reflection (element ln of v and of u, u zero in the lanes >= N; rh = tred_rcp(h))
template <int N>
__device__ __forceinline__ void tred_reflect(double &v, double u, double rh) {
    const double vu = wave_sum_n<N>(u * v);
    v -= (vu * rh) * u;
}

// REFLECT: apply the reflections to v as they are formed (v <- Q' v); else v is left alone (tred_qtv does it afterwards,
// from the stored reflectors, with the same operations)
template <int I, bool REFLECT>
__device__ __forceinline__ void tred_reg_steps(double (&a)[NF], double *__restrict__ A, double &ev, double &hvv,
                                               double &v, double &td, int ln) {
    if constexpr (I >= 1) {
        constexpr int l = I - 1;
        const bool act = ln <= l;
        if (ln == I) td = a[I];                              // row I and column I are final from here on
        // (the selects are pinned to their step: left alone the compiler defers all forty to the end of the function and
        // keeps their lane masks in scalar registers until then -- 80 of them, spilled through v_writelane)
        asm volatile("" : "+v"(td));
        const double x = act ? a[I] : 0.0;                   // A(ln, I) = A(I, ln): row I left of the diagonal
        if constexpr (l == 0) {
            const double f = lane_bcast_u(x, 0);   // (outside the lane-dependent branch: inside it, x is only
            if (ln == I) { ev = f; hvv = 0.0; }    // guaranteed to be computed for the lanes that take the branch)
        } else {
            const double f = lane_bcast_u(x, l);
            const double hoff = wave_sum_n<l>(ln < l ? x * x : 0.0);
            // Nothing left of column l (hoff == 0): already tridiagonal here, no reflection -- e = f, hv = 0.  The step is not
            // branched around: with u = 0 and a harmless h every update below adds an exact zero, and straight-line code
            // lets the register allocator update the columns in place (around a branch it copied every column in every step:
            // 1800 of the reduction's 9900 instructions).
            const bool refl = hoff != 0.0;
            double h = refl ? hoff + f * f : 1.0;
            const double sq = tred_sqrt(h), g = f >= 0 ? -sq : sq;
            h -= f * g;
            const double u = refl ? ((ln == l) ? f - g : x) : 0.0;   // Householder vector (0 beyond l)
            if (act) as_lds(A)[I + LDA * ln] = u;
            if (ln == I) { ev = refl ? g : f; hvv = refl ? h : 0.0; }
            asm volatile("" : "+v"(ev), "+v"(hvv));
            const RowRep U = row_replicate<l / 16 + 1>(u);
            double c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0;   // A u in four chains by k mod 4
            tred_matvec_bc<0, l>(a, U, c0, c1, c2, c3);
            const double rh = tred_rcp(h);
            if constexpr (REFLECT) tred_reflect<I>(v, u, rh);
            // q = p - (u'p / 2h) u with p = A u / h, written as rh (P - (K rh / 2) u), P = A u, K = u'P
            const double P = act ? (c0 + c1) + (c2 + c3) : 0.0;
            const double K = wave_sum_n<I>(P * u);
            const double qv = __builtin_fma(-(K * (0.5 * rh)), u, P) * rh;
            const RowRep Q = row_replicate<l / 16 + 1>(qv);
            tred_update_bc<l>(a, U, Q, -u, -qv);         // (column l first: the next step starts from it)
        }
        tred_reg_steps<I - 1, REFLECT>(a, A, ev, hvv, v, td, ln);
    }
}
template <bool REFLECT>
__device__ __forceinline__ void tred_reg(double *__restrict__ A, double &v, double &td, double &ev, double &hvv, int ln) {
    double a[NF];
    const int row = ln < NF ? ln : 0;                        // lanes >= NF: never active, any finite values do
#pragma unroll
    for (int k = 0; k < NF; ++k) a[k] = as_lds(A)[row + LDA * k];
    wave_sync();                                         // rows of A are overwritten with the u_i below
    td = 0.0; ev = 0.0; hvv = 0.0;                           // e[ln], hv[ln]: each lane keeps its own (lane 0: none)
    tred_reg_steps<NF - 1, REFLECT>(a, A, ev, hvv, v, td, ln);
    if (ln == 0) td = a[0];
    wave_sync();
}
// The reduction alone, for a workgroup whose other wavefronts take the reflections of g (tred_qtv) meanwhile
struct TredOut { double td, te, hv, gt; };
__device__ __noinline__ TredOut tred_only(double *__restrict__ A, const int ln) {
    double td, ev, hvv, v = 0.0;
    tred_reg<false>(A, v, td, ev, hvv, ln);
    return TredOut{td, ev, hvv, 0.0};
}
// Q'v after the fact, by one wavefront: the reflections n-1 ... 2 in turn, each exactly as tred_reg applies it (same
// operands, same operations), from the reflectors in A and hv = |u|^2 / 2 per lane.
template <int I>
__device__ __forceinline__ void tred_qtv_steps(const double (&uu)[NF], double hv_l, double &v) {
    if constexpr (I >= 2) {
        const double h = lane_bcast_u(hv_l, I);
        if (h != 0.0) tred_reflect<I>(v, uu[I], tred_rcp(h));
        tred_qtv_steps<I - 1>(uu, hv_l, v);
    }
}
__device__ __noinline__ double tred_qtv(double *__restrict__ A, double hv_l, double v, const int ln) {
    double uu[NF];
#pragma unroll
    for (int i = 2; i < NF; ++i) uu[i] = ln < i ? as_lds(A)[i + LDA * ln] : 0.0;
    tred_qtv_steps<NF - 1>(uu, hv_l, v);
    return v;
}

// ---------------------------------------------------------------------------------------------------------
// Full symmetric eigen-decomposition by one wavefront: tred_wave, accumulation of the transformation,
// implicit-shift QL.  On exit w = eigenvalues (unordered), column i of A = eigenvector i.  Lane j owns row j
// (QL) or column j (accumulation); every lane carries the scalar recurrences redundantly, so all control flow
// is wave-uniform and the QL phase needs no barrier.
// ---------------------------------------------------------------------------------------------------------
__device__ inline void eig_sym_wave(double *A, double *w, double *e, double *q, int ln) {
    tred_wave(A, w, e, q, ln);
    for (int i = 1; i < NF; ++i) {   // column i above the diagonal: u_i / h_i, as the accumulation expects
        const double h = w[i];
        if (h != 0.0 && ln < i) A[ln + LDA * i] = A[i + LDA * ln] / h;
    }
    wave_sync();
    for (int i = 0; i < NF; ++i) {
        const int l = i - 1;
        const double hi = w[i];
        if (l >= 0 && hi != 0.0 && ln <= l) {
            double g = 0.0;
            for (int k = 0; k <= l; ++k) g += A[i + LDA * k] * A[k + LDA * ln];
            for (int k = 0; k <= l; ++k) A[k + LDA * ln] -= g * A[k + LDA * i];
        }
        wave_sync();
        const double dii = A[i + LDA * i];
        wave_sync();
        w[i] = dii;
        if (ln == 0) A[i + LDA * i] = 1.0;
        if (ln <= l) { A[ln + LDA * i] = 0.0; A[i + LDA * ln] = 0.0; }
        wave_sync();
    }
    // implicit-shift QL on (w, e); rotations are applied to row ln of the eigenvector matrix
    {
        const double ev = (ln >= 1 && ln < NF) ? e[ln] : 0.0;
        wave_sync();
        if (ln >= 1 && ln < NF) e[ln - 1] = ev;
        if (ln == 0) e[NF - 1] = 0.0;
        wave_sync();
    }
    const bool row = ln < NF;
    for (int l = 0; l < NF; ++l) {
        for (int iter = 0; iter < 60; ++iter) {
            int m = l;
            for (; m < NF - 1; ++m) {
                const double dd = fabs(w[m]) + fabs(w[m + 1]);
                if (fabs(e[m]) + dd == dd) break;
            }
            if (m == l) break;
            double g = (w[l + 1] - w[l]) / (2.0 * e[l]);
            double r = sqrt(g * g + 1.0);
            g = w[m] - w[l] + e[l] / (g + (g >= 0 ? r : -r));
            double s = 1.0, c = 1.0, p = 0.0;
            bool underflow = false;
            double zc = row ? A[ln + LDA * m] : 0.0;        // z(ln, col), carried from one rotation to the next
            int col = m;
            for (int i = m - 1; i >= l; --i) {
                const double ei = e[i];
                const double f = s * ei, b = c * ei;
                const double r2 = f * f + g * g;
                if (r2 == 0.0) { e[i + 1] = 0.0; w[i + 1] -= p; e[m] = 0.0; underflow = true; break; }
                const double ir = 1.0 / sqrt(r2);
                e[i + 1] = r2 * ir;
                s = f * ir; c = g * ir;
                g = w[i + 1] - p;
                r = (w[i] - g) * s + 2.0 * c * b;
                p = s * r;
                w[i + 1] = g + p;
                g = c * r - b;
                if (row) {
                    const double zi = A[ln + LDA * i];
                    A[ln + LDA * (i + 1)] = s * zi + c * zc;
                    zc = c * zi - s * zc;
                }
                col = i;
            }
            if (row) A[ln + LDA * col] = zc;
            if (underflow) continue;
            w[l] -= p; e[l] = g; e[m] = 0.0;
        }
    }
    wave_sync();
}

// ---------------------------------------------------------------------------------------------------------
// Tridiagonal-space pieces of the trust-region solve.  td / te: diagonal and sub-diagonal (te[k] couples k-1, k).
// The serial recurrences are carried redundantly by every lane (uniform values, identical LDS writes).
// ---------------------------------------------------------------------------------------------------------
// Both serial recurrences below run on the characteristic polynomials P_k of the leading k x k blocks
// (P_{k+1} = (d_k - x) P_k - e_k^2 P_{k-1}) instead of on their ratios: the chain then holds a multiplication and an
// FMA per step, no division.  The pair (P_{k-1}, P_k) is rescaled by a power of two -- exactly, so nothing below
// depends on when -- every POLY_PERIOD steps while the matrix entries are within 2^+-200 of one (a step changes the
// magnitude by at most the norm of T), else after every step (`wide`).
// The vectors of the tridiagonal problem live one element per lane (td_l = T(ln, ln), te_l = T(ln - 1, ln), ...).
// The serial chains are carried by every lane redundantly; element k reaches them through a scalar register pair
// (v_readlane, k static), not through LDS: a broadcast LDS read in front of every other step of a chain is a
// ~100-cycle wait the compiler would not hoist (17 k cycles per secular iteration, measured), and the element of a
// result that lane k needs is kept by lane k (no 64-lanes-to-one-address stores).
#ifndef POLY_PERIOD
#define POLY_PERIOD 4
#endif
__device__ __forceinline__ void poly_rescale(double &pm, double &pc) {
    const int ex = __builtin_amdgcn_frexp_exp(fmax(fabs(pc), fabs(pm)));
    pm = __builtin_ldexp(pm, -ex); pc = __builtin_ldexp(pc, -ex);
}
__device__ __forceinline__ bool poly_wide_range(double norm_bound) { return !(norm_bound > 1e-60 && norm_bound < 1e60); }
// LDL' of T + lam I (positive definite by construction of lam): ip_l = 1 / pivot ln = P_ln / P_{ln+1} (one division
// per lane, in parallel), mk_l = multiplier ln
// (NE2: -te^2 row-replicated -- it does not change from one shift to the next, the caller forms it once; the result carries
// 1 / pivot and the multipliers both per lane and row-replicated, for the sweeps below)
struct TriFac { double ip_l, mk_l; RowRep IP, NMK; };
__device__ __forceinline__ TriFac tri_factor(double td_l, double te_l, double te2_l, const RowRep &NE2, double lam, bool wide, int ln) {
    double pm = 1.0, pc = lane_bcast_u(td_l, 0) + lam;
    double pa = pm, pb = pc;                  // lane k: P_k and P_{k+1} at one scale
    if (!wide) {
        const RowRep TDL = row_replicate<4>(td_l + lam);
        static_for<1, NF>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            const double pn = __builtin_fma(bcast_mov<k>(TDL), pc, bcast_mov<k>(NE2) * pm);
            if (ln == k) { pa = pc; pb = pn; }
            pm = pc; pc = pn;
            if constexpr (k % POLY_PERIOD == 0) poly_rescale(pm, pc);
        });
    } else {
#pragma unroll 1
        for (int k = 1; k < NF; ++k) {
            const double pn = __builtin_fma(lane_bcast_u(td_l, k) + lam, pc, -lane_bcast_u(te2_l, k) * pm);
            if (ln == k) { pa = pc; pb = pn; }
            pm = pc; pc = pn;
            poly_rescale(pm, pc);
        }
    }
    TriFac F;
    F.ip_l = pa / pb;
    const double ip_prev = __shfl_up(F.ip_l, 1, 64);
    F.mk_l = ln >= 1 ? te_l * ip_prev : 0.0;
    F.IP = row_replicate<4>(F.ip_l); F.NMK = row_replicate<4>(-F.mk_l);
    return F;
}
// y = (T + lam I)^-1 rhs; in: rhs_l = element ln of the right-hand side, returns element ln of y.  The forward
// sweep stays in registers.
// NTE: -te row-replicated (the caller's, like NE2); RHS: the right-hand side row-replicated
__device__ __forceinline__ double tri_solve(const RowRep &NTE, const TriFac &F, const RowRep &RHS, int ln) {
    double r[NF];
    double prev = bcast_mov<0>(RHS);
    r[0] = prev;
    static_for<1, NF>([&](auto kc) {          // prev = rhs_k - mk_k prev
        constexpr int k = decltype(kc)::value;
        double t = bcast_mov<k>(RHS);
        fmac_bcast<k>(t, F.NMK, prev);
        prev = t; r[k] = t;
    });
    double yn = prev * bcast_mov<NF - 1>(F.IP);
    double mine = ln == NF - 1 ? yn : 0.0;
    static_for_down<NF - 2, 0>([&](auto kc) { // y_k = (r_k - te_{k+1} y_{k+1}) / pivot_k
        constexpr int k = decltype(kc)::value;
        double t = r[k];
        fmac_bcast<k + 1>(t, NTE, yn);
        yn = t * bcast_mov<k>(F.IP);
        if (ln == k) mine = yn;
    });
    return mine;
}
__device__ __forceinline__ double tri_solve(const RowRep &NTE, const TriFac &F, double rhs_l, int ln) {
    return tri_solve(NTE, F, row_replicate<4>(rhs_l), ln);
}
// y' (T + lam I)^-1 y = |D^-1/2 L^-1 y|^2 from the factorisation: one forward sweep, no back substitution
__device__ __forceinline__ double tri_quad(const TriFac &F, double y_l) {
    const RowRep Y = row_replicate<4>(y_l);
    double w = bcast_mov<0>(Y);
    double q0 = w * w * bcast_mov<0>(F.IP), q1 = 0.0;
    static_for<1, NF>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        double t = bcast_mov<k>(Y);
        fmac_bcast<k>(t, F.NMK, w);           // w = y_k - mk_k w
        w = t;
        const double ww = w * w;
        if constexpr (k & 1) fmac_bcast<k>(q1, F.IP, ww); else fmac_bcast<k>(q0, F.IP, ww);
    });
    return q0 + q1;
}
// Sturm count #{eigenvalues of T < x} = number of sign changes in P_0 .. P_n; an exact zero takes the sign opposite to
// its predecessor's.  The unrolled form only notes that a zero occurred and leaves the count to the careful loop.
__device__ __noinline__ int sturm_count_careful(const double *__restrict__ td, const double *__restrict__ te2, double x) {
    double pm = 1.0, pc = td[0] - x;
    if (pc == 0.0) pc = -1e-300;
    bool negp = pc < 0;
    int c = negp;
#pragma unroll 1
    for (int k = 1; k < NF; ++k) {
        double pn = __builtin_fma(td[k] - x, pc, -te2[k] * pm);
        if (pn == 0.0) pn = pc > 0 ? -1e-300 : 1e-300;
        pm = pc; pc = pn;
        poly_rescale(pm, pc);
        const bool neg = pc < 0;
        c += neg != negp;
        negp = neg;
    }
    return c;
}
__device__ __forceinline__ int sturm_count(const double *__restrict__ td, const double *__restrict__ te2, double x,
                                           bool wide) {
    if (!wide) {
        double pm = 1.0, pc = td[0] - x;
        bool zero = pc == 0.0;
        bool negp = pc < 0;
        int c = negp;
#pragma unroll
        for (int k = 1; k < NF; ++k) {
            const double pn = __builtin_fma(td[k] - x, pc, -te2[k] * pm);
            zero |= pn == 0.0;
            pm = pc; pc = pn;
            if (k % POLY_PERIOD == 0) poly_rescale(pm, pc);
            const bool neg = pn < 0;
            c += neg != negp;
            negp = neg;
        }
        if (__ballot(zero) == 0ull) return c;
    }
    return sturm_count_careful(td, te2, x);
}
// Smallest (thr = 1) or largest (thr = NF) eigenvalue of T inside [a, b] by 64-way multisection on the Sturm count
__device__ inline double tri_extreme(const double *__restrict__ td, const double *__restrict__ te2, double a, double b,
                                     int thr, int passes, int ln, bool wide, double &lower) {
    for (int pass = 0; pass < passes; ++pass) {
        const double x = a + (b - a) * ((double)(ln + 1) * (1.0 / 65.0));
        const int c = sturm_count(td, te2, x, wide);
        const unsigned long long mask = __ballot(c >= thr);
        double na, nb;
        if (mask) {
            const int jb = __ffsll((long long)mask) - 1;
            nb = lane_bcast(x, jb);
            const double below = lane_bcast(x, jb > 0 ? jb - 1 : 0);
            na = jb > 0 ? below : a;
        } else { na = lane_bcast(x, 63); nb = b; }
        a = na; b = nb;
        if (b - a <= 8.881784197001252e-16 * fmax(fabs(a), fabs(b))) break;
    }
    lower = a;
    return 0.5 * (a + b);
}

// Both extreme eigenvalues in one go, several shifts per lane.  A Sturm pass is a chain of 41 dependent steps whose
// latency (~70 cycles a step: multiply -> FMA -> every few steps a rescale) leaves the SIMD idle more than half of the time,
// and whose T(k, k), T(k - 1, k)^2 operands the compiler would not keep out of the chain (one exposed LDS read per two
// steps: 24 us for the 12 + 3 passes of tri_extreme, measured; holding the 82 operands in registers instead made the
// function spill inside the loop: 35 us).  Here T stays one element per lane and reaches the chains through scalar registers
// (v_readlane, four per step shared by all chains: no memory access and no register array), and every lane runs
// SEVERAL independent recurrences side by side: STURM_M shifts of the smallest eigenvalue's bracket and, during the first
// WMAX_PASSES passes, one shift of the largest's -- whose three passes so cost next to nothing.  (STURM_M = 2, a 129-way
// multisection in 11 passes, issues more instructions per step than the chain's latency hides: 1 is the measured optimum
// of the estimate 3 x 98 vs 2 x 70 cycles per step.)  Shift i of a bracket sits in lane i / M, chain i % M; x_i = a +
// (b - a) (i + 1) / (64 M + 1) is evaluated by the lanes and by the bracket update with the same expression.  narrow: |T|
// within 2^+-60 of one -- the pair (P_{k-1}, P_k) then needs its exact power-of-two rescaling only every 16th step.
#ifndef STURM_IMPL
#define STURM_IMPL 1       // 0: tri_extreme (one chain, T read from LDS), 1: tri_extremes
#endif
#ifndef STURM_M
#define STURM_M 1
#endif
#define WMIN_PASSES (STURM_M == 1 ? 12 : 11)
#define WMAX_PASSES 3
template <int NCH, int PERIOD>
__device__ __forceinline__ bool sturm_counts(const RowRep &TD, const RowRep &E2, const double (&x)[NCH], int (&c)[NCH]) {
    // Per chain and step: subtract, multiply, FMA for the recurrence; the sign changes are collected as bits -- the XOR of
    // the sign bits of P_k and P_{k-1}, shifted into an accumulator by one v_alignbit -- and counted by two popcounts at
    // the end; |P_k| is folded into a running minimum whose being zero sends the whole pass to the careful loop (an exact
    // zero of a P_k needs its sign convention, and two in a row would break the rescaling).  Comparing and adding booleans
    // cost twice the instructions (34 VALU per two-chain step, issue-bound at 148 cycles: measured).
    double pm[NCH], pc[NCH], tiny[NCH];
    unsigned hp[NCH], acc0[NCH], acc1[NCH];
    const double td0 = bcast_mov<0>(TD);
#pragma unroll
    for (int m = 0; m < NCH; ++m) {
        pm[m] = 1.0; pc[m] = td0 - x[m]; tiny[m] = fabs(pc[m]);
        hp[m] = (unsigned)__double2hiint(pc[m]);
        acc0[m] = hp[m] >> 31; acc1[m] = 0u;          // P_0 = 1 is positive
    }
    static_for<1, NF>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        const double tdk = bcast_mov<k>(TD), e2k = bcast_mov<k>(E2);
#pragma unroll
        for (int m = 0; m < NCH; ++m) {
            const double pn = __builtin_fma(tdk - x[m], pc[m], -e2k * pm[m]);
            tiny[m] = fmin(tiny[m], fabs(pn));
            const unsigned h = (unsigned)__double2hiint(pn);
            if (k < 32) acc0[m] = __builtin_amdgcn_alignbit(acc0[m], h ^ hp[m], 31);
            else acc1[m] = __builtin_amdgcn_alignbit(acc1[m], h ^ hp[m], 31);
            hp[m] = h;
            pm[m] = pc[m]; pc[m] = pn;
            if (k % PERIOD == 0) poly_rescale(pm[m], pc[m]);
        }
    });
    bool zero = false;
#pragma unroll
    for (int m = 0; m < NCH; ++m) { c[m] = __popc(acc0[m]) + __popc(acc1[m]); zero |= tiny[m] == 0.0; }
    return zero;
}
__device__ __forceinline__ double sturm_shift(double a, double b, int i, int ways) { return a + (b - a) * ((double)(i + 1) * (1.0 / (double)(ways + 1))); }
// bracket update: i0 = first shift (in index order) whose count reached the threshold, -1: none
__device__ __forceinline__ void sturm_narrow(double &a, double &b, int i0, int ways) {
    if (i0 < 0) a = sturm_shift(a, b, ways - 1, ways);
    else { const double nb = sturm_shift(a, b, i0, ways), na = i0 > 0 ? sturm_shift(a, b, i0 - 1, ways) : a; a = na; b = nb; }
}
// guess (NaN: none): the smallest eigenvalue of the matrix this one grew out of -- the previous accepted point's.  The
// first pass then puts its 64 shifts around it, four times further out from one to the next (16 ulp ... the whole
// Gershgorin interval), instead of spreading them evenly: near convergence the bracket it leaves is a few ulp-decades
// wide instead of 1/65 of the interval, which saves about three of the twelve passes.  Any guess gives a valid bracket
// (the counts decide); a poor one costs at most one pass.
// DO_MIN / DO_MAX: which of the two searches this call runs.  Both (one wavefront does it all): the largest eigenvalue's
// three passes ride along as a second chain.  A fused workgroup gives the two searches to two wavefronts; they do not depend
// on each other -- a pass that meets an exact zero recounts all its chains with the careful loop, whose counts are the fast
// path's whenever that one is valid -- so the brackets, and the results, are the same bit for bit.
template <bool DO_MIN, bool DO_MAX>
__device__ __forceinline__ void tri_extremes(const double *__restrict__ td, const double *__restrict__ te2, double lo, double hi,
                                             int ln, bool wide, bool narrow, double &wmin, double &wmin_lower, double &wmax,
                                             double guess = __builtin_nan("")) {
    static_assert(STURM_M == 1, "one shift of the smallest eigenvalue's bracket per lane");
    const double td_l = td[ln < NF ? ln : 0], e2_l = te2[ln < NF ? ln : 0];
    const RowRep TD = row_replicate<4>(td_l), E2 = row_replicate<4>(e2_l);   // T for the chains of every pass (DPP broadcasts)
    double a = lo, b = hi, a2 = lo, b2 = hi;        // brackets of the smallest / the largest eigenvalue
    constexpr int W = 64;
    bool done1 = false;
    for (int pass = 0; pass < (DO_MIN ? WMIN_PASSES : WMAX_PASSES) && !done1; ++pass) {
        const bool with_max = DO_MAX && pass < WMAX_PASSES;
        double x[2];
        int c[2] = {0, 0};
        x[0] = sturm_shift(a, b, ln, W);
        const bool warm = DO_MIN && pass == 0 && guess > lo && guess < hi;   // (false for a NaN)
        if (warm) {
            const double d0 = fmax(fabs(guess), fmax(fabs(lo), fabs(hi)) * 9.313225746154785e-10) * 3.552713678800501e-15;   // 2^-30, 2^-48
            double xw;
            if (ln < 32) xw = guess - __builtin_ldexp(d0, 2 * (31 - ln));
            else if (ln == 32) xw = guess;
            else xw = guess + __builtin_ldexp(d0, 2 * (ln - 33));
            x[0] = fmin(fmax(xw, lo), hi);
        }
        x[1] = sturm_shift(a2, b2, ln, W);
        bool zero;
        if (wide) zero = true;
        else if (DO_MIN && with_max) zero = narrow ? sturm_counts<2, 16>(TD, E2, x, c) : sturm_counts<2, POLY_PERIOD>(TD, E2, x, c);
        else {   // one chain: the smallest eigenvalue's once the largest's bracket is final (a wavefront alone on its SIMD is
                 // issue-bound: an idle second chain is not free), or the largest's alone
            double x1[1] = {DO_MIN ? x[0] : x[1]};
            int c1[1];
            zero = narrow ? sturm_counts<1, 16>(TD, E2, x1, c1) : sturm_counts<1, POLY_PERIOD>(TD, E2, x1, c1);
            c[DO_MIN ? 0 : 1] = c1[0];
        }
        if (wide || __ballot(zero) != 0ull) {       // an exact zero of some P_k, or entries far from one: the careful loop
            if (DO_MIN) c[0] = sturm_count_careful(td, te2, x[0]);
            if (with_max) c[1] = sturm_count_careful(td, te2, x[1]);
        }
        if constexpr (DO_MIN) {
            const unsigned long long mask = __ballot(c[0] >= 1);
            const int i0 = mask ? __ffsll((long long)mask) - 1 : -1;
            if (warm) {      // the shifts were not evenly spaced: the neighbours of the first one that counted an eigenvalue
                if (i0 < 0) a = __shfl(x[0], 63, 64);
                else { const double nb = __shfl(x[0], i0, 64), na = i0 > 0 ? __shfl(x[0], i0 - 1, 64) : a; a = na; b = nb; }
            } else sturm_narrow(a, b, i0, W);
        }
        if (with_max) {
            const unsigned long long mask = __ballot(c[1] >= NF);
            sturm_narrow(a2, b2, mask ? __ffsll((long long)mask) - 1 : -1, W);
        }
        if constexpr (DO_MIN) done1 = b - a <= 8.881784197001252e-16 * fmax(fabs(a), fabs(b)) && !(pass < WMAX_PASSES);
    }
    if constexpr (DO_MIN) { wmin_lower = a; wmin = 0.5 * (a + b); }
    if constexpr (DO_MAX) wmax = 0.5 * (a2 + b2);
}

// diagnostics: sub-problems solved as interior Newton steps / on the boundary / hard case, total and maximum
// number of secular-equation iterations (celeste_optim_stats)
__device__ unsigned long long g_optim_stats[5];
#ifdef OPTIM_TIMING   // debug builds (tools/variants): shader-clock cycles per section of the step kernel, lane 0
__device__ unsigned long long g_optim_clk[16];
#define OPT_TICK(k) do { const long long now__ = clock64(); if (threadIdx.x == 0) atomicAdd(&g_optim_clk[k], (unsigned long long)(now__ - tick__)); tick__ = clock64(); } while (0)
#define OPT_TICK_DECL long long tick__ = clock64()
#else
#define OPT_TICK(k) do { } while (0)
#define OPT_TICK_DECL do { } while (0)
#endif

#ifdef OPTIM_DEBUG_T
__device__ double *g_dbg_T;   // debug builds: (td, te, Q'g, hv) of every sub-problem, 4 x NF doubles per workgroup
#endif
#define TRI_MAXC 4   // largest cluster of lowest eigenvalues the hard-case test handles in the tridiagonal basis
struct TriLds { double *A, *hv, *td, *te, *te2, *q; };   // hv, te, q: tred_wave's (OPTIM_TRED_LDS builds only)

// Result of a sub-problem solve, per lane: element ln of the step, the model decrease, the interior flag, and whether the
// tridiagonal-space solver solved it (false: the hard case it hands to the eigen-decomposition).
struct TrResult { double p, m; int interior, solved; };

// Trust-region step in the tridiagonal basis, in two parts so that a REJECTED step -- same Hessian, same gradient, smaller
// radius (19 % of the iterations of a joint-inference run) -- repeats only the second:
//   tri_reduce   In: L.A = H (destroyed), g (lane register).  T = Q' H Q, Q' g, the extreme eigenvalues of T.  Leaves
//                the reflectors in L.A and (td, te^2) in LDS; everything else is in the TriForm it returns (per lane:
//                td, te, hv, gt; uniform: wmin, wmax, wmin_lower, norm_bound).
//   tri_step     In: that state and delta.  Out: step p (lane register), model decrease m, interior flag; solved = 0 in
//                the hard case (caller falls back to the eigen-decomposition).
// NOT inlined: 22 k instructions between them, and their register allocation (248 VGPRs at the edge of the two-waves-
// per-SIMD budget) should not depend on the kernel around them -- optim_step_kernel, tr_solve_kernel and
// optim_fused_kernel call the same code, so their steps agree bit for bit by construction.
struct TriForm { double td, te, hv, gt, wmin, wmax, wmin_lower, norm_bound; };
// second half of tri_reduce: T (one element per lane) -> LDS copy for the Sturm counts, Gershgorin interval, extreme eigenvalues
// (DO_MIN / DO_MAX: tri_extremes)
template <bool DO_MIN, bool DO_MAX>
__device__ __forceinline__ TriForm tri_reduce_tail(TriLds L, double td_l, double te_l, double hv_l, double gt_l, int ln, double wmin_guess) {
    const bool fr = ln < NF;
    OPT_TICK_DECL;
    const double te2_l = te_l * te_l;
#ifdef OPTIM_DEBUG_T
    if (DO_MIN && fr && g_dbg_T) { double *o = g_dbg_T + (size_t)blockIdx.x * 4 * NF; o[ln] = td_l; o[NF + ln] = te_l; o[2 * NF + ln] = gt_l; o[3 * NF + ln] = hv_l; }
#endif
    if constexpr (DO_MIN) {   // (the other wavefront of a split search finds them there)
        if (fr) { L.td[ln] = td_l; L.te2[ln] = te2_l; }   // the Sturm passes read T from LDS (hoisted: loop invariant)
        wave_sync();
    }
    OPT_TICK(4);
    // Gershgorin interval, extreme eigenvalues
    double wmin = 0.0, wmax = 0.0, wmin_lower = 0.0, norm_bound;
    {
        const double te_next = __shfl_down(te_l, 1, 64);
        const double ea = fabs(te_l), eb = (ln + 1 < NF) ? fabs(te_next) : 0.0;
        double lo = fr ? td_l - ea - eb : INFINITY, hi = fr ? td_l + ea + eb : -INFINITY;
        for (int o = 32; o >= 1; o >>= 1) { lo = fmin(lo, __shfl_xor(lo, o, 64)); hi = fmax(hi, __shfl_xor(hi, o, 64)); }
        const double pad = 4.440892098500626e-16 * fmax(fabs(lo), fabs(hi)) + 1e-300;
        lo -= pad; hi += pad;
        norm_bound = fmax(fabs(lo), fabs(hi));
        const bool wide = poly_wide_range(norm_bound);
        const bool narrow = norm_bound > 8.673617379884035e-19 && norm_bound < 1.152921504606847e18;   // 2^-60 .. 2^60
        tri_extremes<DO_MIN, DO_MAX>(L.td, L.te2, lo, hi, ln, wide, narrow, wmin, wmin_lower, wmax, wmin_guess);
    }
    OPT_TICK(5);
    return TriForm{td_l, te_l, hv_l, gt_l, wmin, wmax, wmin_lower, norm_bound};
}
__device__ __noinline__ TriForm tri_reduce(TriLds L, double g, int ln, double wmin_guess = __builtin_nan("")) {
    const bool fr = ln < NF;
    OPT_TICK_DECL;
    // T = Q' H Q and gt = Q' g (reflections n-1 ... 2 in turn) in one pass
    double gt_l = fr ? g : 0.0, td_l, te_l, hv_l;
#if OPTIM_TRED_LDS
    tred_wave(L.A, L.hv, L.te, L.q, ln);
    td_l = fr ? L.A[ln + LDA * ln] : 0.0; te_l = fr ? L.te[ln] : 0.0; hv_l = fr ? L.hv[ln] : 0.0;
    for (int i = NF - 1; i >= 2; --i) {
        const double h = L.hv[i];
        if (h == 0.0) continue;
        const double u = ln < i ? L.A[i + LDA * ln] : 0.0;
        gt_l -= (wave_sum_dpp(u * gt_l) / h) * u;
    }
#else
    tred_reg<true>(L.A, gt_l, td_l, te_l, hv_l, ln);
#endif
    OPT_TICK(3);
    return tri_reduce_tail<true, true>(L, td_l, te_l, hv_l, gt_l, ln, wmin_guess);
}
// tri_reduce's second half on its own, split over two wavefronts of a fused workgroup (tred_only and tred_qtv are the first
// half): the smallest eigenvalue by the calling wavefront, which wrote T to LDS ...
__device__ __noinline__ TriForm tri_spectrum(TriLds L, TredOut T, int ln, double wmin_guess) {
    return tri_reduce_tail<true, false>(L, T.td, T.te, T.hv, T.gt, ln, wmin_guess);
}
// ... and the largest by another one (T from LDS: L.td, L.te -- the sub-diagonal itself, not its square -- behind a barrier)
__device__ __noinline__ double tri_spectrum_max(TriLds L, int ln) {
    const bool fr = ln < NF;
    return tri_reduce_tail<false, true>(L, fr ? L.td[ln] : 0.0, fr ? L.te[ln] : 0.0, 0.0, 0.0, ln, __builtin_nan("")).wmax;
}

// p = Q y: the reflections 2 ... n-1 in turn; reflection i has its vector in the lanes < i
template <int I>
__device__ __forceinline__ void tri_qy_steps(const double (&uu)[NF], double rhv_l, double &y) {
    if constexpr (I < NF) {
        const double rh = lane_bcast_u(rhv_l, I);
        if (rh != 0.0) y -= (wave_sum_n<I>(uu[I] * y) * rh) * uu[I];
        tri_qy_steps<I + 1>(uu, rhv_l, y);
    }
}
__device__ __noinline__ TrResult tri_step(TriLds L, TriForm TF, double delta, int ln, int secular_iters) {
    double p_out = 0.0, m_out = 0.0;
    int interior_out = 0;
    const bool fr = ln < NF;
    OPT_TICK_DECL;
    const double td_l = TF.td, te_l = TF.te, hv_l = TF.hv, gt_l = TF.gt;
    const double wmin = TF.wmin, wmax = TF.wmax, wmin_lower = TF.wmin_lower, norm_bound = TF.norm_bound;
    const double te2_l = te_l * te_l;
    const bool wide = poly_wide_range(norm_bound);
    if (fr) { L.td[ln] = td_l; L.te2[ln] = te2_l; }   // (the hard-case test counts eigenvalues of T from LDS; a repeated step comes here without tri_reduce)
    wave_sync();
    const double d2 = delta * delta;
    int interior = 0;
    double y = 0.0;
    // row-replicated for the DPP broadcasts of the recurrences below: they are the same for every shift
    const RowRep NE2 = row_replicate<4>(-te2_l), NTE = row_replicate<4>(-te_l), NG = row_replicate<4>(-gt_l);
    TriFac F;
    if (wmin >= 1e-8) {
        F = tri_factor(td_l, te_l, te2_l, NE2, 0.0, wide, ln);
        y = tri_solve(NTE, F, NG, ln);
        interior = wave_sum_dpp(y * y) <= d2;
    }
    if (!interior) {
        const double lambda_lb = -wmin + fmax(1e-8, 1e-8 * (wmax - wmin));
        double lambda = lambda_lb;
        bool hard = false;
        // first iterate of the secular equation, at lambda_lb
        F = tri_factor(td_l, te_l, te2_l, NE2, lambda, wide, ln);
        y = tri_solve(NTE, F, NG, ln);
        double q2 = wave_sum_dpp(y * y);
        bool fresh = true;                                       // (F, y, q2) belong to `lambda`
        // Hard-case candidate: g orthogonal (1e-10) to the eigenvectors of every eigenvalue within 1e-10 of the
        // smallest, AND the step at lambda_lb without its components along them no longer than delta.  Those
        // components are at most 1e-10 / (lambda_lb + wmin) each, so a step at lambda_lb that is longer than delta by
        // more than that cannot be a hard case: the common situation, screened here without any eigenvector.
        const double gap = lambda_lb + wmin;
        if (wmin < 0 && (!OPTIM_HARD_SCREEN || q2 <= d2 * (1.0 + 1e-6) + TRI_MAXC * (1e-10 / gap) * (1e-10 / gap))) {
            // Their number is a Sturm count; an orthonormal basis z_0 .. z_{mc-1} of their span comes from inverse
            // iteration with the shift just below the smallest eigenvalue (the Sturm count at wmin_lower is 0, so
            // T - shift I is positive definite) and Gram-Schmidt.
            int mc = sturm_count(L.td, L.te2, wmin + 1e-10, wide);
            if (mc < 1) mc = 1;
            if (mc > TRI_MAXC) { if (ln == 0) atomicAdd(&g_optim_stats[2], 1ull); return TrResult{0.0, 0.0, 0, 0}; }
            const double shift = wmin_lower - 4.440892098500626e-16 * norm_bound;
            const TriFac Fs = tri_factor(td_l, te_l, te2_l, NE2, -shift, wide, ln);
            bool orth = true;
            double zc[TRI_MAXC];
            for (int j = 0; j < mc && orth; ++j) {
                double z = fr ? 1.0 + 0.5 * sin(1.7 * ln + 0.3 + 2.1 * j) : 0.0;
                for (int it = 0; it < 4; ++it) {
                    z = tri_solve(NTE, Fs, z, ln);
#pragma unroll
                    for (int jj = 0; jj < TRI_MAXC; ++jj)
                        if (jj < j) z -= wave_sum_dpp(z * zc[jj]) * zc[jj];
                    z *= 1.0 / sqrt(wave_sum_dpp(z * z));
                }
                if (fabs(wave_sum_dpp(z * gt_l)) > 1e-10) orth = false;
#pragma unroll
                for (int jj = 0; jj < TRI_MAXC; ++jj) if (jj == j) zc[jj] = z;
            }
            if (orth) {
                double yh = y;
#pragma unroll
                for (int j = 0; j < TRI_MAXC; ++j)
                    if (j < mc) yh -= wave_sum_dpp(yh * zc[j]) * zc[j];
                const double p2 = wave_sum_dpp(yh * yh);
                if (p2 <= d2) {   // N&W (4.45): to the boundary along the lowest eigenvector
                    hard = true;
                    y = yh + sqrt(d2 - p2) * zc[0];
                    if (ln == 0) atomicAdd(&g_optim_stats[2], 1ull);
                }
            }
        }
        if (!hard) {
        int it = 0;
        for (; it < secular_iters; ++it) {
            if (!fresh) {
                F = tri_factor(td_l, te_l, te2_l, NE2, lambda, wide, ln);
                y = tri_solve(NTE, F, NG, ln);
                q2 = wave_sum_dpp(y * y);
            }
            fresh = false;
#if OPTIM_Q3_SOLVE
            const double q3 = wave_sum_dpp(y * tri_solve(NTE, F, y, ln));
#else
            const double q3 = tri_quad(F, y);                    // y' (T + lambda)^-1 y
#endif
            const double prev = lambda;
            lambda += q2 * (sqrt(q2) - delta) / (delta * q3);
            if (lambda < lambda_lb) lambda = 0.5 * (prev - lambda_lb) + lambda_lb;
            if (fabs(lambda - prev) < 1e-10 || lambda <= prev) break;
        }
        if (ln == 0) {
            atomicAdd(&g_optim_stats[1], 1ull); atomicAdd(&g_optim_stats[3], (unsigned long long)it);
            atomicMax(&g_optim_stats[4], (unsigned long long)it);
        }
        }
    }
    OPT_TICK(6);
    // model value g'p + p'Hp / 2 in the tridiagonal basis
    {
        const double y_prev = __shfl_up(y, 1, 64), y_next = __shfl_down(y, 1, 64), te_next = __shfl_down(te_l, 1, 64);
        double ty = td_l * y;
        if (ln > 0) ty += te_l * y_prev;
        if (ln + 1 < NF) ty += te_next * y_next;
        m_out = wave_sum_dpp(fr ? gt_l * y + 0.5 * y * ty : 0.0);
    }
    // p = Q y: reflections 2 ... n-1 in turn (the vectors are fetched first: the chain of wave sums then runs
    // without an LDS wait per reflection)
    {
        double uu[NF];
#pragma unroll
        for (int i = 2; i < NF; ++i) uu[i] = ln < i ? as_lds(L.A)[i + LDA * ln] : 0.0;
        const double rhv_l = hv_l != 0.0 ? 1.0 / hv_l : 0.0;   // one division per lane, in parallel, instead of one in every link of the chain
        tri_qy_steps<2>(uu, rhv_l, y);
    }
    OPT_TICK(7);
    if (interior && ln == 0) atomicAdd(&g_optim_stats[0], 1ull);
    p_out = y;
    interior_out = interior;
    return TrResult{p_out, m_out, interior_out, 1};
}

// both parts in a row
__device__ __forceinline__ TrResult tri_tr_solve(TriLds L, double g, double delta, int ln, int secular_iters) {
    return tri_step(L, tri_reduce(L, g, ln), delta, ln, secular_iters);
}

// The same sub-problem through the full eigen-decomposition (Optim.jl's own route): the hard-case fallback of
// tri_tr_solve, and OptParams.solver = 1.  In: A = H (LDS, destroyed), g, delta.  w, e, q, cv: NF doubles of LDS each.
__device__ __noinline__ TrResult eig_tr_solve(double *A, double *w, double *e, double *q, double *cvs, double g, double delta_in,
                                              int tid, int secular_iters) {
    double step, m;
    int interior;
    const bool fr = tid < NF;
    double *const s_g = cvs;   // g until its projections are formed, the step's coefficients afterwards
    if (fr) s_g[tid] = g;
    eig_sym_wave(A, w, e, q, tid);
    double qg = 0.0;
    if (fr) for (int k = 0; k < NF; ++k) qg += A[k + LDA * tid] * s_g[k];
    const double wi = fr ? w[tid] : 0.0;
    // extreme eigenvalues (and the lane of the smallest)
    double wmin = fr ? wi : INFINITY, wmax = fr ? wi : -INFINITY;
    int imin = tid;
    for (int o = 32; o >= 1; o >>= 1) {
        const double om = __shfl_xor(wmin, o, 64);
        const int oi = __shfl_xor(imin, o, 64);
        if (om < wmin || (om == wmin && oi < imin)) { wmin = om; imin = oi; }
        wmax = fmax(wmax, __shfl_xor(wmax, o, 64));
    }
    const double delta = delta_in, d2 = delta * delta;
    interior = 0;
    if (wmin >= 1e-8) {
        const double r = fr ? qg / wi : 0.0;
        interior = wave_sum(r * r) <= d2;
    }
    double cv;
    if (interior) cv = fr ? -qg / wi : 0.0;
    else {
        const double lambda_lb = -wmin + fmax(1e-8, 1e-8 * (wmax - wmin));
        double lambda = lambda_lb;
        bool hard = false;
        cv = 0.0;
        if (wmin < 0) {
            const bool low = fr && fabs(wi - wmin) <= 1e-10;          // eigenvalues tied with the smallest
            if (__ballot(low && fabs(qg) > 1e-10) == 0ull) {          // g orthogonal to all their eigenvectors
                const double r = (fr && !low) ? qg / (wi + lambda) : 0.0;
                const double p2 = wave_sum(r * r);
                if (p2 <= d2) {   // N&W (4.45): to the boundary along the lowest eigenvector
                    hard = true;
                    cv = tid == imin ? sqrt(d2 - p2) : -r;
                }
            }
        }
        if (!hard) {
            for (int it = 0; it < secular_iters; ++it) {
                cv = fr ? -qg / (wi + lambda) : 0.0;
                const double q2 = wave_sum(cv * cv), q3 = wave_sum(fr ? cv * cv / (wi + lambda) : 0.0);
                const double prev = lambda;
                lambda += q2 * (sqrt(q2) - delta) / (delta * q3);
                if (lambda < lambda_lb) lambda = 0.5 * (prev - lambda_lb) + lambda_lb;
                if (fabs(lambda - prev) < 1e-10 || lambda <= prev) break;
            }
        }
    }
    m = wave_sum(fr ? qg * cv + 0.5 * wi * cv * cv : 0.0);
    wave_sync();
    if (fr) cvs[tid] = cv;
    wave_sync();
    step = 0.0;
    if (fr) for (int i = 0; i < NF; ++i) step += A[tid + LDA * i] * cvs[i];
    return TrResult{step, m, interior, 1};
}

// to_bound! by one wavefront: x (41, LDS) -> vs (44).  COH: vs is read by other workgroups of the same launch (stc)
// tmp: 27 doubles of LDS scratch (the simplex groups' exponentials and sums)
template <bool COH = false, class OP>
__device__ inline void to_bound_wave(const double *x, const double *pos0, const OP &op, double *vs, int ln, double *tmp) {
    if (ln < 26) {
        double lo, hi, sc;
        box_bounds(ln, pos0, op, lo, hi, sc);
        stc<COH>(vs + ln, (1.0 / (1.0 + exp(-x[ln] / sc))) * (hi - lo) + lo);
    } else if (ln >= 32 && ln < 56) {
        const int g = (ln - 32) >> 3, i = (ln - 32) & 7;
        const double p = simplex_prob_lane(x, g, i, tmp, tmp + 24);
        const int n = c_simplex_n[g];
        const double lo = c_simplex_lo[g];
        if (i < n) stc<COH>(vs + c_simplex_b0[g] + i, (1 - n * lo) * p + lo);
    }
}

// ---------------------------------------------------------------------------------------------------------
// One Newton trust-region iteration of ONE target by the calling workgroup of NTHR threads: optim_step_kernel (NTHR = 64:
// a workgroup is a wavefront) and optim_fused_kernel (NTHR = 256) run the same body.  With several wavefronts the chain
// rule and the copies are shared out -- every entry is still computed by one thread with the same arithmetic, so the
// result does not depend on NTHR -- and the trust-region sub-problem stays with wavefront 0.
// COH: the optimiser state, the saved Hessian and the target's row of vp move between workgroups inside one launch
// (ldc / stc).  In: h = 44 x 44 bound-space Hessian, ev_d = 44-gradient, ft_in = -elbo, st_in = status of the
// evaluation at the trial point S.xt.  Returns 1 when the target is finished (its row of vp holds the result), else 0
// (its row holds the next trial point).
// ---------------------------------------------------------------------------------------------------------
struct StepShared {
    // LDS budget of optim_step_kernel: 19.7 KB per workgroup so that 8 workgroups (2 waves per SIMD, the VGPR limit) fit a
    // CU and a batch of 2000 targets is resident in one round.
    double sA[LDA * NF];       // rows 0..43: H J (44 x 41) -> J'HJ (41 x 41, negated) -> solver; rows 41..44 spare
    double sx[NF], sg[NF], sw[NF], se[NF], scv[NF];
    double sgt_q[NF];          // trial gradient during the chain rule, Householder scratch afterwards
    double sU[9 * NF];         // chain-rule tables, then the solver's vectors (disjoint lifetimes)
    int s_flag[2];             // 0: accept, 1: done
    double s_delta;            // trust-region radius after the update
    int s_iter;                // iteration counter after the update (the tag of a step computed ahead, fused_speculate)
};

template <bool COH, int NTHR, class OP>
__device__ __forceinline__ int optim_step_target(StepShared &Z, const int tid, OptState &S, double *__restrict__ Hs, double *__restrict__ vp_row,
                                                 const double *__restrict__ h, const double *__restrict__ ev_d, double ft_in,
                                                 int st_in, const OP &op, double *__restrict__ Ts = nullptr,
                                                 const double *__restrict__ Sp = nullptr) {
    // Ts (optional, TRI_STATE doubles per target): the reduced form of the last accepted point's sub-problem (tri_reduce),
    // so that a rejected step -- same Hessian and gradient, smaller radius -- goes straight to tri_step
    // Sp (optional, SPEC_STATE doubles per target): the quarter-radius step from that point, computed ahead of time by the
    // fused kernel while the trial point was being evaluated (fused_speculate): trial point, model decrease, interior flag,
    // and the iteration it belongs to
    constexpr int PARTS = NTHR / 64;
    double *const sA = Z.sA, *const sx = Z.sx, *const sg = Z.sg, *const sw = Z.sw, *const se = Z.se, *const scv = Z.scv;
    double *const sU = Z.sU;
    double *const sgt = Z.sgt_q, *const sq = Z.sgt_q;
    double *const sd = sU, *const sJb = sU + 44, *const sHb = sU + 70;                   // 44 + 26 + 26
    double (*const sp)[8] = reinterpret_cast<double (*)[8]>(sU + 96);                    // 3 x 8
    double (*const sJs)[8][7] = reinterpret_cast<double (*)[8][7]>(sU + 120);            // 3 x 8 x 7 (ends at 288)
    double *const std_ = sU, *const ste2 = sU + NF;
    int *const s_flag = Z.s_flag;

    OPT_TICK_DECL;
    const int ln = PARTS == 1 ? tid : (tid & 63), part = PARTS == 1 ? 0 : (tid >> 6);

    // scalars of the optimiser state, loaded now so that their latency hides behind the chain rule below
    const double S_f = ldc<COH>(&S.f), S_m = ldc<COH>(&S.m), S_delta = ldc<COH>(&S.delta);
    const int S_iter = ldc<COH>(&S.iter), S_interior = ldc<COH>(&S.interior), S_evals = ldc<COH>(&S.evals);
    // the smallest eigenvalue at the last accepted point (NaN before the first: the buffer is preset), for tri_extremes
    const double wmin_prev = Ts ? ldc<COH>(Ts + NF * NF + 4 * NF) : __builtin_nan("");

    // Is the step rejected?  Only the value decides (rho, below), and every thread can tell from the scalars it holds.
    // A rejected point's gradient and Hessian are never used: the chain rule is skipped.
    bool rejected = false;
    if (st_in == CELESTE_OK && S_iter >= 0) {
        double rho;
        if (fabs(S_m) <= 2.220446049250313e-16) rho = 1.0;
        else if (S_m > 0) rho = 0.25 - 1.0;
        else rho = (S_f - ft_in) / (0 - S_m);
        rejected = !(rho > 0.1);
    }
    if (!rejected) {
    // ---- chain rule to the free parameters at the evaluated point xt (propagate_derivatives!) ----
    if (tid < CEL_P) sd[tid] = ev_d[tid];
    if (tid < NF) sx[tid] = ldc<COH>(&S.xt[tid]);
    __syncthreads();
    if (tid < 26) {
        double lo, hi, sc;
        box_bounds(tid, S.pos0, op, lo, hi, sc);
        const double s = 1.0 / (1.0 + exp(-sx[tid] / sc)), w = hi - lo;
        sJb[tid] = w * s * (1 - s) / sc;
        sHb[tid] = w * s * (1 - s) * (1 - 2 * s) / (sc * sc);
    } else if (tid >= 32 && tid < 56) {   // the simplex groups, eight lanes each (simplex_prob_lane)
        const int g = (tid - 32) >> 3, i = (tid - 32) & 7;
        const double p = simplex_prob_lane(sx, g, i, &sp[0][0], &sJs[0][0][0]);
        if (i < c_simplex_n[g]) sp[g][i] = p;
    }
    __syncthreads();
    OPT_TICK(9);
    // simplex Jacobians d bound_{b0+a} / d free_{f0+j} = (1 - n lo) p_a ((a == j) - p_j)
    for (int k = tid; k < 3 * 56; k += NTHR) {
        const int g = k / 56, r = k - g * 56, a = r / 7, j = r - a * 7;
        const int n = c_simplex_n[g];
        sJs[g][a][j] = (a < n && j < n - 1) ? (1 - n * c_simplex_lo[g]) * sp[g][a] * ((a == j) - sp[g][j]) : 0.0;
    }
    __syncthreads();
    if (tid < NF) {   // gradient: J' d
        double s;
        if (tid < 26) s = sJb[tid] * sd[tid];
        else {
            const int g = tid < 27 ? 0 : (tid < 34 ? 1 : 2), j = tid - c_simplex_f0[g];
            s = 0;
            for (int a = 0; a < c_simplex_n[g]; ++a) s += sJs[g][a][j] * sd[c_simplex_b0[g] + a];
        }
        sgt[tid] = -s;  // minimise -elbo
    }
    OPT_TICK(10);
    // J' H J, J block diagonal (26 scalars + simplex blocks 2x1, 8x7, 8x7): rows first (lane = row; the columns are
    // shared out over the wavefronts) ...
    if (ln < CEL_P) {   // row `ln` of the bound-space Hessian (lanes read consecutive addresses)
        for (int i = part; i < 26; i += PARTS) sA[ln + LDA * i] = h[ln + CEL_P * i] * sJb[i];
        for (int g = 0; g < 3; ++g) {
            const int n = c_simplex_n[g], b0 = c_simplex_b0[g], f0 = c_simplex_f0[g];
            double hv[8];
#pragma unroll
            for (int b = 0; b < 8; ++b) hv[b] = b < n ? h[ln + CEL_P * (b0 + b)] : 0.0;
            for (int j = part; j < n - 1; j += PARTS) {
                double o = 0;
#pragma unroll
                for (int b = 0; b < 8; ++b) o += hv[b] * sJs[g][b][j];
                sA[ln + LDA * (f0 + j)] = o;
            }
        }
    }
    __syncthreads();
    OPT_TICK(11);
    // ... then columns (lane = column; the rows are shared out over the wavefronts), plus the second derivatives of the
    // transform contracted with the bound gradient, negated (minimise -elbo)
    {
        // the lane's column is fetched in one batch of independent LDS reads and worked on in registers (entry by
        // entry through LDS, with loop bounds the compiler could not see, this pass was a chain of ~300 exposed LDS
        // latencies: 7.2 us of the kernel's 95)
        double *col = sA + LDA * (ln < NF ? ln : 0);
        double cv[CEL_P];
        auto fetch = [&]() {
#pragma unroll
            for (int i = 0; i < CEL_P; ++i) cv[i] = col[i];
        };
        auto columns = [&]() {
#pragma unroll
            for (int i = 0; i < 26; ++i) {
                if (PARTS > 1 && i % PARTS != part) continue;
                double o = -(cv[i] * sJb[i]);
                if (i == ln) o -= sd[i] * sHb[i];
                col[i] = o;
            }
            constexpr int GN[3] = {2, 8, 8}, GB0[3] = {26, 28, 36}, GF0[3] = {26, 27, 34};
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                constexpr double lo_[3] = {0.005, 0.01 / 8, 0.01 / 8};
                const int n = GN[g], b0 = GB0[g], f0 = GF0[g];
                const int kk = ln - f0;
                const bool own = kk >= 0 && kk < n - 1;
                const double *pp = sp[g];
                const double scl = 1 - n * lo_[g];
#pragma unroll
                for (int jj = 0; jj < 7; ++jj) {
                    if (jj >= n - 1) break;
                    if (PARTS > 1 && (f0 + jj) % PARTS != part) continue;
                    double o = 0;
#pragma unroll
                    for (int a = 0; a < 8; ++a) if (a < n) o += sJs[g][a][jj] * cv[b0 + a];
                    if (own) {
                        for (int a = 0; a < n; ++a) {
                            const double d2 = pp[a] * (((a == jj) - pp[jj]) * ((a == kk) - pp[kk]) - pp[jj] * ((jj == kk) - pp[kk]));
                            o += sd[b0 + a] * scl * d2;
                        }
                    }
                    col[f0 + jj] = -o;
                }
            }
        };
        if constexpr (PARTS == 1) {
            if (ln < NF) { fetch(); columns(); }
        } else {
            fetch();
            __syncthreads();   // every wavefront has its copy of the column before any entry of it is replaced
            if (ln < NF) columns();
        }
    }
    __syncthreads();
    OPT_TICK(12);
    if (ln < NF) for (int i = part; i < ln; i += PARTS) sA[ln + LDA * i] = sA[i + LDA * ln];   // exactly symmetric
    __syncthreads();
    }

    OPT_TICK(0);
    // ---- accept / reject, radius update, convergence (N&W Alg. 4.1 as in Optim.jl's NewtonTrustRegion) ----
    if (part == 0) {
        const double dxl = tid < NF ? fabs(sx[tid] - ldc<COH>(&S.x[tid])) : 0.0, gl = tid < NF ? fabs(sgt[tid]) : 0.0;
        double dx = dxl, gmax = gl;
        for (int o = 32; o >= 1; o >>= 1) { dx = fmax(dx, __shfl_xor(dx, o, 64)); gmax = fmax(gmax, __shfl_xor(gmax, o, 64)); }
        if (tid == 0) {
            const double ft = ft_in;
            int accept = 1, done = 0;
            double delta = S_delta;
            stc<COH>(&S.evals, S_evals + 1);
            if (st_in != CELESTE_OK) { stc<COH>(&S.status, st_in); accept = 0; done = 1; }
            else if (S_iter >= 0) {
                const double m = S_m;
                double rho;
                if (fabs(m) <= 2.220446049250313e-16) rho = 1.0;
                else if (m > 0) rho = 0.25 - 1.0;
                else rho = (S_f - ft) / (0 - m);
                if (rho < 0.25) delta *= 0.25;
                else if (rho > 0.75 && !S_interior) delta = fmin(2 * delta, op.delta_hat);
                accept = rho > 0.1;
                if (accept && (dx <= op.xtol_abs || fabs(ft - S_f) <= op.ftol_rel * fabs(ft) || gmax <= op.gtol)) done = 1;
            } else if (gmax <= op.gtol) done = 1;   // already stationary at the starting point: no iteration at all
            if (accept) stc<COH>(&S.f, ft);
            stc<COH>(&S.delta, delta); Z.s_delta = delta;
            stc<COH>(&S.iter, S_iter + 1); Z.s_iter = S_iter + 1;
            if (S_iter + 1 >= op.max_iters) done = 1;
            s_flag[0] = accept; s_flag[1] = done;
        }
    }
    __syncthreads();
    const int accept = s_flag[0], done = s_flag[1];
    bool spec = false;
#ifdef OPTIM_TIMING
    if (!accept && tid == 0) atomicAdd(&g_optim_clk[13], 2400ull);   // (counted as 1 us per rejected step in tools/gpu_optim_sections.py)
#endif
    if (accept) {
        if (tid < NF) { stc<COH>(&S.x[tid], sx[tid]); stc<COH>(&S.g[tid], sgt[tid]); sg[tid] = sgt[tid]; }
        if (!done) for (int k = tid; k < NF * NF; k += NTHR) { const int j = k / NF; stc<COH>(&Hs[k], sA[(k - j * NF) + LDA * j]); }
    } else {
        if (tid < NF) { sx[tid] = ldc<COH>(&S.x[tid]); sg[tid] = ldc<COH>(&S.g[tid]); }
        // was this very step computed ahead of time?  (the tag is the iteration it was computed for; stale ones never match)
        if (Sp) Sp += (S_iter & 1) * SPEC_STATE;
        spec = Sp && !done && op.solver != 1 && ldc<COH>(Sp + NF + 2) == (double)S_iter;
        // the sub-problem's matrix: its reduced form if it was kept (reflectors; tri_step needs nothing else of H), else H
        const double *const src = (Ts && op.solver != 1) ? Ts : Hs;
        if (!done && !spec) for (int k = tid; k < NF * NF; k += NTHR) { const int j = k / NF; sA[(k - j * NF) + LDA * j] = ldc<COH>(&src[k]); }
    }
    __syncthreads();
    OPT_TICK(1);
    if (done) {
        if (tid == 0) stc<COH>(&S.done, 1);
        to_bound_wave<COH>(sx, S.pos0, op, vp_row, tid, sU);
        return 1;
    }

    // ---- trust-region sub-problem at the accepted point (N&W section 4.3): the tridiagonalisation by all four wavefronts
    // of a large workgroup, the rest by wavefront 0 ----
    // (a repeated step takes the reduced form kept with the target; `spec`, `accept`, the solver: the same for every thread)
    const bool reduce_here = !spec && op.solver != 1 && !(Ts && !accept);
    TriForm TF4 = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if constexpr (PARTS == 4 && OPTIM_TRED_SPLIT) {
        if (reduce_here) {
            // wavefront 0: T = Q'HQ, then its smallest eigenvalue; meanwhile wavefront 1: Q'g from the stored reflectors (the step
            // itself needs it, the eigenvalues do not), wavefront 2: the largest eigenvalue -- the same operations tri_reduce
            // performs in one wavefront, so the same bits
            TredOut T4 = {0.0, 0.0, 0.0, 0.0};
            if (part == 0) {
                OPT_TICK_DECL;
                T4 = tred_only(sA, ln);
                OPT_TICK(3);
                if (ln < NF) { sw[ln] = T4.hv; std_[ln] = T4.td; ste2[ln] = T4.te * T4.te; sq[ln] = T4.te; }
            }
            __syncthreads();
            if (part == 0) TF4 = tri_spectrum({sA, sw, std_, se, ste2, sq}, T4, ln, wmin_prev);
            else if (part == 1) { const double gt = tred_qtv(sA, ln < NF ? sw[ln] : 0.0, ln < NF ? sg[ln] : 0.0, ln); if (ln < NF) se[ln] = gt; }
            else if (part == 2) { const double wmax = tri_spectrum_max({sA, sw, std_, sq, ste2, sq}, ln); if (ln == 0) scv[0] = wmax; }
            __syncthreads();
            if (part == 0) { TF4.gt = ln < NF ? se[ln] : 0.0; TF4.wmax = scv[0]; }
        }
    }
    if (part == 0) {
        const bool fr = tid < NF;
        TrResult R = {0.0, 0.0, 0, 0};
        if (spec) {
            // the step the rejection calls for is already there: trial point (taken as it is below), model decrease, interior flag
            R.m = ldc<COH>(Sp + NF); R.interior = ldc<COH>(Sp + NF + 1) != 0.0; R.solved = 1;
        } else if (op.solver != 1) {
            const TriLds L = {sA, sw, std_, se, ste2, sq};
            TriForm TF;
            if (Ts && !accept) {
                // a repeated step: the reduced form from the accepted point's solve
                const double *v = Ts + NF * NF;
                TF.td = fr ? ldc<COH>(v + tid) : 0.0; TF.te = fr ? ldc<COH>(v + NF + tid) : 0.0;
                TF.hv = fr ? ldc<COH>(v + 2 * NF + tid) : 0.0; TF.gt = fr ? ldc<COH>(v + 3 * NF + tid) : 0.0;
                TF.wmin = ldc<COH>(v + 4 * NF); TF.wmax = ldc<COH>(v + 4 * NF + 1);
                TF.wmin_lower = ldc<COH>(v + 4 * NF + 2); TF.norm_bound = ldc<COH>(v + 4 * NF + 3);
            } else {
                if constexpr (PARTS == 4 && OPTIM_TRED_SPLIT) TF = TF4;
                else TF = tri_reduce(L, fr ? sg[tid] : 0.0, tid, wmin_prev);
                if (Ts) {   // kept for the steps that may be rejected from here (stores only: nothing waits for them)
                    for (int k = tid; k < NF * NF; k += 64) { const int j = k / NF; stc<COH>(&Ts[k], sA[(k - j * NF) + LDA * j]); }
                    double *v = Ts + NF * NF;
                    if (fr) { stc<COH>(v + tid, TF.td); stc<COH>(v + NF + tid, TF.te); stc<COH>(v + 2 * NF + tid, TF.hv); stc<COH>(v + 3 * NF + tid, TF.gt); }
                    if (tid == 0) { stc<COH>(v + 4 * NF, TF.wmin); stc<COH>(v + 4 * NF + 1, TF.wmax); stc<COH>(v + 4 * NF + 2, TF.wmin_lower); stc<COH>(v + 4 * NF + 3, TF.norm_bound); }
                }
            }
            R = tri_step(L, TF, Z.s_delta, tid, op.secular_iters);
            if (!R.solved) {   // hard case: restore H and diagonalise it
                for (int k = tid; k < NF * NF; k += 64) { const int j = k / NF; sA[(k - j * NF) + LDA * j] = ldc<COH>(&Hs[k]); }
                wave_sync();
            }
        }
        if (!R.solved) R = eig_tr_solve(sA, sw, se, sq, scv, fr ? sg[tid] : 0.0, Z.s_delta, tid, op.secular_iters);
        OPT_TICK(2);   // the whole sub-problem (sections 3-7 are its parts)
        if (tid == 0) { stc<COH>(&S.m, R.m); stc<COH>(&S.interior, R.interior); }
        if (fr) {
            const double xn = spec ? ldc<COH>(Sp + tid) : sx[tid] + R.p;
            stc<COH>(&S.xt[tid], xn); sx[tid] = xn;
        }
    }
    __syncthreads();
    to_bound_wave<COH>(sx, S.pos0, op, vp_row, tid, sU);   // next evaluation point
    OPT_TICK(8);
#ifdef OPTIM_TIMING
    if (tid == 0) atomicAdd(&g_optim_clk[15], 1ull);
#endif
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// optim_step_kernel
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64, 2)   // 2 waves per SIMD: 8 workgroups per CU (see the LDS budget of StepShared)
optim_step_kernel(double *__restrict__ vp, const int32_t *__restrict__ targets, const int32_t *__restrict__ active,
                  const double *__restrict__ ev_v, const double *__restrict__ ev_d, const double *__restrict__ ev_h,
                  const int32_t *__restrict__ ev_status, OptParams op, OptState *__restrict__ st,
                  double *__restrict__ Hstate, int32_t *__restrict__ next_active, int32_t *__restrict__ next_targets,
                  int32_t *__restrict__ next_count, int32_t *__restrict__ live, int32_t *__restrict__ blocks_done,
                  volatile int32_t *__restrict__ host_count, double *__restrict__ Tstate) {
    // The launch's last workgroup to finish publishes the number of targets still running to page-locked host memory
    // (the host sizes later launches by it), clears the counter this launch read (`live`: the launch after next
    // counts into it) and re-arms blocks_done: no memset and no copy between two Newton iterations.
    auto finish = [&]() {
        if (threadIdx.x != 0) return;
        __threadfence();
        if (atomicAdd(blocks_done, 1) == (int)gridDim.x - 1) {
            *host_count = atomicAdd(next_count, 0);
            if (live) *live = 0;
            *blocks_done = 0;
            __threadfence_system();
        }
    };
    if (live && (int)blockIdx.x >= *live) { finish(); return; }   // the grid is sized by an older, larger count
    __shared__ StepShared Z;
    const int li = blockIdx.x;
    const int slot = active[li];
    const int t = targets[slot];
    const int done = optim_step_target<false, 64>(Z, threadIdx.x, st[slot], Hstate + (size_t)slot * NF * NF, vp + (size_t)t * CEL_P,
                                                  ev_h + (size_t)li * CEL_P * CEL_P, ev_d + (size_t)li * CEL_P, -ev_v[li],
                                                  ev_status[li], op, Tstate ? Tstate + (size_t)slot * TRI_STATE : nullptr);
    if (!done && threadIdx.x == 0) {
        const int pos = atomicAdd(next_count, 1);
        next_active[pos] = slot; next_targets[pos] = t;
    }
    finish();
}

// ---------------------------------------------------------------------------------------------------------
// tr_solve_kernel (celeste_tr_solve_batch): the trust-region sub-problem alone, one wavefront per problem -- the
// solver of optim_step_kernel exposed for tests (tests/test_gpu_tr_subproblem.py compare it with the CPU restatement
// and with a 60-digit solution).  solver 0: tridiagonal-space solve with the eigen-decomposition as the hard-case
// fallback, exactly as the optimiser; 1: eigen-decomposition; 2: tridiagonal-space solve only (status 1 = fell back).
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64, 2)
tr_solve_kernel(const double *__restrict__ H, const double *__restrict__ g, const double *__restrict__ delta, int solver,
                int secular_iters, double *__restrict__ p, double *__restrict__ m_out, int32_t *__restrict__ interior_out,
                int32_t *__restrict__ fallback_out) {
    __shared__ double sA[LDA * NF];
    __shared__ double sw[NF], se[NF], scv[NF], sq[NF], sU[2 * NF];
    const int b = blockIdx.x, tid = threadIdx.x;
    const bool fr = tid < NF;
    const double *Hb = H + (size_t)b * NF * NF;
    for (int k = tid; k < NF * NF; k += 64) { const int j = k / NF; sA[(k - j * NF) + LDA * j] = Hb[k]; }
    for (int k = tid; k < (LDA - NF) * NF; k += 64) { const int j = k / (LDA - NF); sA[NF + (k - j * (LDA - NF)) + LDA * j] = 0.0; }
    const double gl = fr ? g[(size_t)b * NF + tid] : 0.0;
    const double dl = delta[b];
    __syncthreads();
    TrResult R = {0.0, 0.0, 0, 0};
    if (solver != 1) {
        const TriLds L = {sA, sw, sU, se, sU + NF, sq};
        R = tri_tr_solve(L, gl, dl, tid, secular_iters);
        if (!R.solved && solver == 0) {
            for (int k = tid; k < NF * NF; k += 64) { const int j = k / NF; sA[(k - j * NF) + LDA * j] = Hb[k]; }
            __syncthreads();
        }
    }
    if (tid == 0) fallback_out[b] = !R.solved && solver != 1;
    if (!R.solved && solver != 2) R = eig_tr_solve(sA, sw, se, sq, scv, gl, dl, tid, secular_iters);
    if (fr) p[(size_t)b * NF + tid] = R.solved ? R.p : 0.0;
    if (tid == 0) { m_out[b] = R.m; interior_out[b] = R.interior; }
}
