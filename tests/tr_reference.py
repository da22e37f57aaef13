"""A 60-digit solution of the optimiser's trust-region sub-problem (test infrastructure).

min g'p + p'Hp/2, |p| <= delta, by the rules of Optim.jl's solve_tr_subproblem! as restated in
oracle/celeste_optim_oracle.c:178-253 (N&W section 4.3): the plain Newton step when the smallest eigenvalue is
>= 1e-8 and the step fits; else lambda_lb = -w_min + max(1e-8, 1e-8 (w_max - w_min)); the hard case when w_min < 0, g is
orthogonal (1e-10) to the eigenvectors within 1e-10 of w_min and the remaining step fits; else the root of
|p(lambda)| = delta to the right of lambda_lb (p(lambda_lb) itself when that is already inside).  Eigen-decomposition
and root in mpmath, so the result is exact to far below fp64 rounding and independent of both solvers under test.
Also: generators of test problems (random, structured with exact zeros, next to the hard case, Celeste's own)."""
import numpy as np


def tr_reference(H, g, delta, dps=60):
    """-> dict(p, lam, kind in {'interior', 'hard', 'boundary', 'lb'}, wmin, wmax, z (lowest eigenvector), mc)"""
    import mpmath as mp
    with mp.workdps(dps):
        n = len(g)
        A = mp.matrix([[mp.mpf(float(H[i, j])) for j in range(n)] for i in range(n)])
        A = (A + A.T) / 2
        E, Q = mp.eigsy(A)
        E = [E[i] for i in range(n)]
        order = sorted(range(n), key=lambda i: E[i])
        gv = mp.matrix([mp.mpf(float(x)) for x in g])
        qg = [sum(Q[k, i] * gv[k] for k in range(n)) for i in range(n)]
        wmin, wmax = E[order[0]], E[order[-1]]
        d2 = mp.mpf(float(delta)) ** 2

        def step(c):
            return np.array([float(sum(Q[k, i] * c[i] for i in range(n))) for k in range(n)])
        z = np.array([float(Q[k, order[0]]) for k in range(n)])
        low = [i for i in range(n) if abs(E[i] - wmin) <= mp.mpf("1e-10")]
        out = dict(wmin=float(wmin), wmax=float(wmax), z=z, mc=len(low))
        if wmin >= mp.mpf("1e-8"):
            c = [-qg[i] / E[i] for i in range(n)]
            if sum(x * x for x in c) <= d2:
                return dict(out, p=step(c), lam=0.0, kind="interior")
        lb = -wmin + max(mp.mpf("1e-8"), mp.mpf("1e-8") * (wmax - wmin))
        if wmin < 0 and all(abs(qg[i]) <= mp.mpf("1e-10") for i in low):
            c = [mp.mpf(0) if i in low else -qg[i] / (E[i] + lb) for i in range(n)]
            p2 = sum(x * x for x in c)
            if p2 <= d2:
                c[order[0]] = mp.sqrt(d2 - p2)
                return dict(out, p=step(c), lam=float(lb), kind="hard")
        phi = lambda lam: sum((qg[i] / (E[i] + lam)) ** 2 for i in range(n)) - d2
        if phi(lb) <= 0:
            lam, kind = lb, "lb"
        else:
            hi = lb + 1
            while phi(hi) > 0:
                hi = lb + 2 * (hi - lb)
            lo = lb
            for _ in range(int(3.4 * dps) + 20):       # bisection: dps digits of the bracket
                mid = (lo + hi) / 2
                if phi(mid) > 0: lo = mid
                else: hi = mid
            lam, kind = (lo + hi) / 2, "boundary"
        c = [-qg[i] / (E[i] + lam) for i in range(n)]
        return dict(out, p=step(c), lam=float(lam), kind=kind)


def error_bound(ref, H, p_norm):
    """what an fp64 solver may differ by from the exact step: rounding in H and g amplified by the conditioning of
    H + lambda I, plus the 1e-10 tolerance of the Newton iteration on lambda"""
    gap = max(ref["wmin"] + ref["lam"], 1e-300)
    scale = max(abs(ref["wmax"]), abs(ref["wmin"])) + abs(ref["lam"])
    return (200 * 2.2e-16 * scale / gap + 2e-10 / gap + 1e-13) * p_norm


def random_problems(rng, n=41):
    """(name, H, g, delta) of increasing nastiness"""
    out = []

    def sym(M): return (M + M.T) / 2
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    # 1-2: positive definite, Celeste's scale (curvatures up to 1e7), small and large radius
    w = 10.0 ** rng.uniform(-2, 7, n)
    H = sym(Q @ np.diag(w) @ Q.T); g = rng.standard_normal(n) * 1e3
    out += [("spd boundary", H, g, 0.01), ("spd interior", H, g, 1e6)]
    # 3-4: indefinite
    w2 = w.copy(); w2[:5] = -10.0 ** rng.uniform(-1, 4, 5)
    H = sym(Q @ np.diag(w2) @ Q.T)
    out += [("indefinite", H, g, 1.0), ("indefinite tiny radius", H, g, 1e-6)]
    # 5: block diagonal with exact zeros (parts already tridiagonal), zero rows
    Hb = np.zeros((n, n))
    for a, b in ((0, 7), (7, 8), (8, 20), (20, 41)):
        B = rng.standard_normal((b - a, b - a)); Hb[a:b, a:b] = sym(B) * 1e3
    Hb[30, :] = 0; Hb[:, 30] = 0; Hb[30, 30] = 5.0
    out += [("blocks with exact zeros", Hb, g, 0.5)]
    # 6: diagonal
    out += [("diagonal", np.diag(rng.uniform(-3, 8, n)), rng.standard_normal(n), 2.0)]
    # 7: exact hard case: g has no component along the lowest eigenvector of a diagonal matrix
    d = rng.uniform(1, 9, n); d[17] = -4.0
    gh = rng.standard_normal(n) * 0.01; gh[17] = 0.0
    out += [("hard case, diagonal", np.diag(d), gh, 3.0)]
    # 8: hard case in a rotated basis (g orthogonal to the lowest eigenvector to rounding)
    wr = rng.uniform(1, 50, n); wr[0] = -2.5
    Hr = sym(Q @ np.diag(wr) @ Q.T); gr = Q[:, 1:] @ (rng.standard_normal(n - 1) * 0.05)
    out += [("hard case, rotated", Hr, gr, 4.0)]
    # 9: next to the hard case: component 1e-7 along the lowest eigenvector, solution near lambda_lb
    out += [("near hard case", Hr, gr + 1e-7 * Q[:, 0], 4.0)]
    # 10: nearly singular positive semi-definite (flat directions, as a star's galaxy parameters)
    wf = 10.0 ** rng.uniform(0, 6, n); wf[:6] = 10.0 ** rng.uniform(-9, -7, 6)
    out += [("flat directions", sym(Q @ np.diag(wf) @ Q.T), rng.standard_normal(n) * 10, 1.0)]
    # 11: double lowest eigenvalue, g orthogonal to both (cluster of 2 in the hard-case test)
    wd = rng.uniform(1, 50, n); wd[0] = wd[1] = -1.0
    out += [("hard case, double eigenvalue", sym(Q @ np.diag(wd) @ Q.T), Q[:, 2:] @ (rng.standard_normal(n - 2) * 0.05), 5.0)]
    # 12: parameters deep in the flat part of their transform: rows and columns scaled by 1e-150, so that their squares
    # leave the exponent range (the reduction's norms underflow to denormals or to zero; its fast square root hands over
    # to the library's there)
    wu = 10.0 ** rng.uniform(0, 5, n)
    D = np.ones(n); D[[3, 19, 33]] = 1e-150
    Hu = sym(Q @ np.diag(wu) @ Q.T) * D[:, None] * D[None, :]
    # (a small radius: the step is then on the boundary with a multiplier far above the three vanishing eigenvalues -- next to
    # them the problem is a hard case whose answer is a matter of tolerances, which is not what this family is about)
    out += [("underflowing rows", sym(Hu), rng.standard_normal(n) * D, 1e-3)]
    return out


def celeste_problems(oracle, scene="star", points=3):
    """(name, H, g, delta): -ELBO's gradient and Hessian in the free parameters at a few iterates of maximize! on a
    sample scene (the matrices the optimiser really sees: curvatures from 1e-3 to 1e7, flat directions)"""
    from celeste_jl_amd import synthetic, cabi
    f = synthetic.make_sample_dataset(scene)
    pb = cabi.Problem(f.images, f.patches, f.neighbors)
    out = []
    vp = f.vp.copy()
    for k in range(points):
        vs0 = vp[0].copy()
        x, vs, _ = oracle.constraints_roundtrip(vs0)
        vpe = vp.copy(); vpe[0] = vs
        v, d, h, _, st = oracle.elbo_one(pb, vpe, 0)
        assert st == 0
        gf, Hf = oracle.propagate(x, vs0, d, h)
        for delta in (1.0, 0.05):
            out.append(("%s iterate %d delta %g" % (scene, 4 * k, delta), -Hf, -gf, delta))
        vp = oracle.maximize(pb, vp, 0, oracle.OptCfg(max_iters=4))[0]
    return out
