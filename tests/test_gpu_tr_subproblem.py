"""The optimiser's trust-region sub-problem on the device (celeste_tr_solve_batch: tridiagonal-space solver and the
eigen-decomposition fallback of optim_step_kernel) against a 60-digit solution of the same rules
(tests/tr_reference.py) and beside the CPU restatement: random, structured, hard-case and Celeste's own Hessians."""
import numpy as np
import pytest

import tr_reference as R

pytestmark = pytest.mark.gpu


def device_errors(problems, oracle):
    from celeste_jl_amd import cabi
    H = np.stack([p[1] for p in problems]); g = np.stack([p[2] for p in problems]); delta = np.array([p[3] for p in problems])
    res = {s: cabi.tr_solve_batch(H, g, delta, solver=s) for s in (0, 1, 2)}
    rows, bad = [], []
    for k, (name, Hk, gk, dk) in enumerate(problems):
        ref = R.tr_reference(Hk, gk, dk)
        pn = max(np.linalg.norm(ref["p"]), 1e-300)
        bound = R.error_bound(ref, Hk, pn)

        def err(p):
            e = np.linalg.norm(p - ref["p"])
            if ref["kind"] == "hard":   # the sign of the lowest eigenvector is free; so is the vector itself in a cluster
                e = min(e, np.linalg.norm(p - ref["p"] + 2 * (ref["z"] @ ref["p"]) * ref["z"]))
                if ref["mc"] > 1:
                    e = abs(np.linalg.norm(p) - pn)
            return e
        cpu = oracle.solve_tr(gk, Hk, dk)
        e = {s: err(res[s][0][k]) for s in (0, 1)}
        fell = int(res[2][3][k])
        if not fell:
            assert np.array_equal(res[2][0][k], res[0][0][k])      # solver 0 = solver 2 unless it fell back
        m_ref = float(gk @ ref["p"] + 0.5 * ref["p"] @ Hk @ ref["p"])
        print("%-34s %-8s gap %.1e | error / bound: tridiagonal %.2e  eigen %.2e  cpu %.2e | fell back %d"
              % (name, ref["kind"], ref["wmin"] + ref["lam"], e[0] / bound, e[1] / bound, err(cpu[0]) / bound, fell))
        for s in (0, 1):
            if int(res[s][2][k]) != (ref["kind"] == "interior"): bad.append((name, s, "interior flag"))
            if not e[s] <= bound: bad.append((name, s, ref["kind"], "step", e[s], bound))
            # the model value g'p + p'Hp/2 the solver reports (rho's denominator); its error follows the step's
            if not abs(res[s][1][k] - m_ref) <= (1e-9 + 10 * bound / pn) * abs(m_ref) + 1e-300:
                bad.append((name, s, "model value", res[s][1][k], m_ref))
        rows.append((ref["kind"], fell))
    assert not bad, bad
    return rows


def test_sub_problem_against_60_digits(oracle):
    rng = np.random.default_rng(11)
    rows = device_errors(R.random_problems(rng), oracle)
    assert {k for k, _ in rows} >= {"interior", "boundary", "hard"}
    assert all(not fell for _, fell in rows)     # clusters of <= 4 lowest eigenvalues stay in the tridiagonal solver


def test_sub_problem_on_celeste_hessians(oracle):
    device_errors(R.celeste_problems(oracle, "star", points=3) + R.celeste_problems(oracle, "galaxy", points=2), oracle)


def test_sub_problem_batch_and_arguments():
    """many problems in one call give what single calls give; bad arguments are refused"""
    from celeste_jl_amd import cabi
    rng = np.random.default_rng(3)
    n = 300
    H = rng.standard_normal((n, 41, 41)); H = (H + H.transpose(0, 2, 1)) * 50
    g = rng.standard_normal((n, 41)); delta = 10.0 ** rng.uniform(-3, 1, n)
    p, m, interior, fell = cabi.tr_solve_batch(H, g, delta)
    assert np.isfinite(p).all() and (np.linalg.norm(p, axis=1) <= delta * (1 + 1e-9)).all()
    for k in (0, 17, 299):
        p1 = cabi.tr_solve_batch(H[k], g[k], delta[k])[0][0]
        assert np.array_equal(p1, p[k])
    model = np.einsum("ni,ni->n", g, p) + 0.5 * np.einsum("ni,nij,nj->n", p, H, p)
    assert np.allclose(model, m, rtol=1e-9, atol=0) and (m < 0).all()
    with pytest.raises(cabi.CelesteError):
        cabi.tr_solve_batch(H[:1], g[:1], delta[:1], solver=3)
