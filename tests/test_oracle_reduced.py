"""The reduced-variable CPU evaluation (oracle/celeste_reduced.c, used only as a timed baseline) against the dense,
reference-faithful restatement: same value, gradient, Hessian, counters."""
import numpy as np
import pytest

from parity_util import assert_parity


@pytest.mark.parametrize("kind", ["star", "galaxy", "two_body", "three_body"])
def test_sample_datasets(oracle, kind):
    from celeste_jl_amd import synthetic, cabi
    f = synthetic.make_sample_dataset(kind)
    pb = cabi.Problem(f.images, f.patches, f.neighbors)
    tg = list(range(len(f.catalog)))
    for flags in (7, 3, 5, 4, 0):
        errs = assert_parity(oracle.reduced_elbo_batch(pb, f.vp, tg, flags, n_threads=1),
                             oracle.elbo_batch(pb, f.vp, tg, flags, n_threads=1), kind)
    print(kind, errs)


def test_crowded_field_with_masks_and_affine_wcs(oracle):
    from celeste_jl_amd import synthetic, cabi
    f = synthetic.make_field(120, 140, 24, seed=11, nan_fraction=0.01)
    f.patches[3][2].active_pixel_bitmap[2:6, :] = False
    J = np.array([[0.95, 0.08], [-0.05, 1.03]])
    for row in f.patches[:6]:
        for p in row:
            p.wcs_jacobian = J.copy()
    pb = cabi.Problem(f.images, f.patches, f.neighbors)
    tg = list(range(24))
    a = oracle.reduced_elbo_batch(pb, f.vp, tg, 7, n_threads=2)
    b = oracle.elbo_batch(pb, f.vp, tg, 7, n_threads=2)
    print(assert_parity(a, b, "reduced vs dense"))
