"""The CPU restatement of the trust-region sub-problem (oracle/celeste_optim_oracle.c:178-253) against a 60-digit
solution of the same rules (tests/tr_reference.py) -- which also validates the reference the GPU test uses."""
import numpy as np
import pytest

import tr_reference as R


def check(name, solve, H, g, delta):
    ref = R.tr_reference(H, g, delta)
    p, interior = solve(H, g, delta)
    pn = np.linalg.norm(ref["p"])
    err = np.linalg.norm(p - ref["p"])
    if ref["kind"] == "hard":       # the sign of the lowest eigenvector is free; so is the vector itself in a cluster
        zc = ref["z"] @ (p - ref["p"])
        err = min(err, np.linalg.norm(p - ref["p"] + 2 * (ref["z"] @ ref["p"]) * ref["z"]))
        if ref["mc"] > 1:
            err = abs(np.linalg.norm(p) - pn)
    bound = R.error_bound(ref, H, max(pn, 1e-300))
    assert interior == (ref["kind"] == "interior"), (name, ref["kind"])
    assert err <= bound, (name, ref["kind"], err, bound)
    return err / bound, ref["kind"]


@pytest.mark.parametrize("seed", [11, 12])
def test_oracle_sub_problem_against_60_digits(oracle, seed):
    rng = np.random.default_rng(seed)
    kinds = set()
    for name, H, g, delta in R.random_problems(rng):
        ratio, kind = check(name, lambda H, g, d: oracle.solve_tr(g, H, d)[::2], H, g, delta)
        kinds.add(kind)
    assert kinds == {"interior", "boundary", "hard"} or kinds == {"interior", "boundary", "hard", "lb"}


@pytest.mark.parametrize("scene", ["star", "galaxy"])
def test_oracle_sub_problem_on_celeste_hessians(oracle, scene):
    for name, H, g, delta in R.celeste_problems(oracle, scene, points=2):
        check(name, lambda H, g, d: oracle.solve_tr(g, H, d)[::2], H, g, delta)
