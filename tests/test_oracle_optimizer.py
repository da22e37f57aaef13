"""CPU: the optimiser restatement (oracle/celeste_optim_oracle.c): constraint transforms, derivative
propagation, eigen-solver, trust-region sub-problem, and the recovery tests of test/test_optimization.jl."""
import math

import numpy as np
import pytest


def test_constraint_roundtrip_and_jacobian(oracle):
    """to_free! / to_bound! round trip (test_constraints.jl) and the analytic Jacobian vs autograd"""
    import torch
    from celeste_jl_amd import synthetic
    vs = synthetic.make_sample_dataset("galaxy").vp[0]
    x, back, J = oracle.constraints_roundtrip(vs, loc_width=1.0)
    assert x.shape == (41,) and np.abs(back - vs).max() < 1e-13
    lo = np.array([vs[0] - 1, vs[1] - 1, 1e-2, 1e-2, -10, .1] + [-1] * 2 + [1e-4] * 2 + [-10] * 8 + [1e-4] * 8)
    hi = np.array([vs[0] + 1, vs[1] + 1, .99, .99, 10, 70] + [10] * 2 + [.1] * 2 + [10] * 8 + [1.0] * 8)

    def to_bound(xf):
        box = torch.sigmoid(xf[:26]) * torch.tensor(hi - lo) + torch.tensor(lo)
        outs = [box]
        for f0, n, l in ((26, 2, 0.005), (27, 8, 0.01 / 8), (34, 8, 0.01 / 8)):
            z = torch.cat([xf[f0:f0 + n - 1], torch.zeros(1, dtype=torch.float64)])
            outs.append((1 - n * l) * torch.softmax(z, 0) + l)
        return torch.cat(outs)

    xt = torch.tensor(x, dtype=torch.float64)
    assert np.abs(to_bound(xt).numpy() - vs).max() < 1e-13
    Jt = torch.autograd.functional.jacobian(to_bound, xt).numpy()
    assert np.abs(J - Jt).max() < 1e-13
    # propagate_derivatives!: free gradient / Hessian of a random quadratic in the bound parameters
    rng = np.random.default_rng(0)
    d = rng.normal(size=44); h = rng.normal(size=(44, 44)); h = h + h.T
    gf, Hf = oracle.propagate(x, vs, d, h, loc_width=1.0)

    def fq(xf):
        b = to_bound(xf) - torch.tensor(vs)
        return torch.tensor(d) @ b + 0.5 * b @ torch.tensor(h) @ b

    gt = torch.autograd.functional.jacobian(fq, xt).numpy()
    Ht = torch.autograd.functional.hessian(fq, xt).numpy()
    assert np.abs(gf - gt).max() <= 1e-12 * np.abs(gt).max()
    assert np.abs(Hf - Ht).max() <= 1e-12 * np.abs(Ht).max()


def test_enforce_clamps_into_the_open_box(oracle):
    """enforce! (ConstraintTransforms.jl:225-253)"""
    from celeste_jl_amd import generic_init_source
    vs = generic_init_source([5.0, 6.0])
    vs[3] = 1.5; vs[8] = 0.5; vs[26:28] = [1.2, -0.1]
    x, back, _ = oracle.constraints_roundtrip(vs)
    assert np.all(np.isfinite(x))
    assert back[3] < 0.99 and back[3] > 0.98 and back[8] < 0.10
    assert abs(back[26:28].sum() - 1.0) < 1e-9 and back[27] >= 0.005


def test_jacobi_eigensolver(oracle):
    rng = np.random.default_rng(1)
    A = rng.normal(size=(41, 41)); A = A + A.T
    w, V = oracle.jacobi_eig(A)
    assert np.abs(w - np.linalg.eigvalsh(A)).max() < 1e-11
    assert np.abs(V @ np.diag(w) @ V.T - A).max() < 1e-11 and np.abs(V.T @ V - np.eye(41)).max() < 1e-12


def test_trust_region_subproblem_kkt(oracle):
    """N&W Theorem 4.1: (H + lambda I) s = -g, lambda >= 0, lambda (delta - |s|) = 0, H + lambda I psd"""
    rng = np.random.default_rng(2)
    for trial in range(6):
        A = rng.normal(size=(41, 41)); H = A + A.T if trial % 2 else A @ A.T + 0.1 * np.eye(41)
        g = rng.normal(size=41)
        for delta in (1e-2, 1.0, 1e2):
            s, m, interior = oracle.solve_tr(g, H, delta)
            assert np.linalg.norm(s) <= delta * (1 + 1e-7)
            assert m == pytest.approx(g @ s + 0.5 * s @ H @ s, rel=1e-9, abs=1e-9)
            if interior:
                assert np.abs(H @ s + g).max() < 1e-8
            else:
                lam = -(s @ (H @ s + g)) / (s @ s)
                assert lam > -1e-8 and np.abs((H + lam * np.eye(41)) @ s + g).max() < 1e-6 * max(1, np.abs(g).max())
                assert np.linalg.eigvalsh(H)[0] + lam > -1e-7
                assert abs(np.linalg.norm(s) - delta) <= 1e-7 * delta
    # hard case: gradient orthogonal to the eigenvector of the smallest (negative) eigenvalue
    H = np.diag(np.concatenate([[-1.0], np.linspace(1, 3, 40)])); g = np.zeros(41); g[1:] = 0.01
    s, m, interior = oracle.solve_tr(g, H, 1.0)
    assert not interior and abs(np.linalg.norm(s) - 1.0) < 1e-9 and m < 0


def _verify_sample_galaxy(vs, pos):
    """test/test_optimization.jl:10-32"""
    from celeste_jl_amd import ids
    from celeste_jl_amd.synthetic import SAMPLE_GALAXY_FLUXES as gf
    assert vs[ids.is_star[1]] >= 0.99
    assert abs(vs[0] - pos[0]) <= 0.1 and abs(vs[1] - pos[1]) <= 0.1
    assert abs(vs[ids.gal_axis_ratio] - 0.7) <= 0.05
    assert abs(vs[ids.gal_frac_dev] - 0.1) <= 0.08
    assert abs(vs[ids.gal_radius_px] - 4.0) <= 0.2
    phi = vs[ids.gal_angle]; phi -= math.floor(phi / math.pi) * math.pi
    assert abs(phi - math.pi / 4) <= 5 * math.pi / 180
    assert abs(math.exp(vs[ids.flux_loc[1]] + 0.5 * vs[ids.flux_scale[1]]) / gf[2] - 1.0) <= 0.05
    true_colors = np.log(gf[1:] / gf[:-1])
    assert np.all(np.abs(vs[ids.color_mean[:, 1]] - true_colors) <= 0.2)


def test_galaxy_optimization(oracle):
    """test_optimization.jl:54-59 (include_kl = false, loc_width = 3.0)"""
    from celeste_jl_amd import synthetic, cabi
    f = synthetic.make_sample_dataset("galaxy")
    pb = cabi.Problem(f.images, f.patches, f.neighbors)
    vp, it, evals, elbo, st = oracle.maximize(pb, f.vp, 0, oracle.OptCfg(loc_width=3.0, include_kl=False))
    assert st == 0 and evals == it + 1
    assert elbo > oracle.elbo_one(pb, f.vp, 0, 0)[0]
    _verify_sample_galaxy(vp[0], [8.5, 9.6])


def test_full_elbo_optimization(oracle):
    """test_optimization.jl:62-68 (KL on, loc_width = 1.0, x_tol = 0)"""
    from celeste_jl_amd import synthetic, cabi
    f = synthetic.make_sample_dataset("galaxy")
    pb = cabi.Problem(f.images, f.patches, f.neighbors)
    vp, it, evals, elbo, st = oracle.maximize(pb, f.vp, 0, oracle.OptCfg(loc_width=1.0, xtol_abs=0.0))
    assert st == 0
    _verify_sample_galaxy(vp[0], [8.5, 9.6])


def test_only_the_active_source_moves(oracle):
    """test_optimization.jl:36-51"""
    from celeste_jl_amd import synthetic, cabi
    f = synthetic.make_sample_dataset("three_body")
    pb = cabi.Problem(f.images, f.patches, f.neighbors)
    vp, it, evals, elbo, st = oracle.maximize(pb, f.vp, 1, oracle.OptCfg(loc_width=1.0, include_kl=False, max_iters=8))
    assert st == 0 and not np.array_equal(vp[1], f.vp[1])
    assert np.array_equal(vp[0], f.vp[0]) and np.array_equal(vp[2], f.vp[2])
