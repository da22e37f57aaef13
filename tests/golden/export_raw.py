"""Raw little-endian export of golden fixtures for tools/reference_golden.jl (Julia 0.6 + Celeste.jl).

usage: python tests/golden/export_raw.py [name ...]      (default: the three sample scenes, whose exports are committed)

Writes tests/golden/raw/<name>.bin (all arrays back to back, little-endian, matrices COLUMN-major as Julia stores them)
and tests/golden/raw/<name>.txt, one line per array: `name dtype ndims dim1 [dim2 ...] byte_offset`.  Only inputs travel:
images (pixels, sky, per-row calibration, band, PSF mixture, the 51 x 51 psfmap stamp), the catalog and the variational
parameters.  Patches, neighbours and ElboArgs are built on the Julia side by the reference's own constructors
(Model.get_sky_patches, Model.find_neighbors, ElboArgs) -- so a reference run pins the patch construction as well.
Fixtures with a constant PSF map and an identity WCS only (the sample scenes and make_field's constant template)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import golden_util as gu  # noqa: E402

DEFAULT = ["sample_star", "sample_galaxy", "sample_two_body"]


def export(name):
    z = np.load(gu.path(name))
    f = gu.arrays_to_field(z)
    arrays = []

    def add(key, a, dtype):
        a = np.asarray(a, dtype=dtype)
        arrays.append((key, a, np.dtype(dtype).newbyteorder("<")))
    N, S = len(f.images), len(f.catalog)
    add("n_images", [N], np.int64)
    add("n_sources", [S], np.int64)
    for n, im in enumerate(f.images):
        assert np.array_equal(im.wcs_jacobian, np.eye(2)) and not im.wcs_world0.any() and not im.wcs_pix0.any(), "identity WCS only"
        assert type(im.psfmap).__name__ == "ConstantPSFMap", "constant PSF map only"
        add("band_%d" % (n + 1), [im.b], np.int64)
        add("pixels_%d" % (n + 1), im.pixels, np.float32)                  # H x W
        add("sky_%d" % (n + 1), im.sky, np.float32)                        # H x W
        add("nelec_per_nmgy_%d" % (n + 1), im.nelec_per_nmgy, np.float32)  # H
        add("psf_%d" % (n + 1), im.psf, np.float64)                        # K x 6: alphaBar, xiBar(2), tauBar 11, 12, 22
        add("psf_stamp_%d" % (n + 1), im.psfmap.stamp, np.float64)         # 51 x 51
    add("pos", [c.pos for c in f.catalog], np.float64)                     # S x 2
    add("is_star", [int(c.is_star) for c in f.catalog], np.int64)
    add("star_fluxes", [c.star_fluxes for c in f.catalog], np.float64)     # S x 5
    add("gal_fluxes", [c.gal_fluxes for c in f.catalog], np.float64)
    add("gal_shape", [[c.gal_frac_dev, c.gal_axis_ratio, c.gal_angle, c.gal_radius_px] for c in f.catalog], np.float64)   # S x 4
    add("vp", f.vp, np.float64)                                            # S x 44
    # For callers that do NOT rebuild the patches themselves (tests/cabi_caller.c, a C program compiled against
    # include/celeste_mi355x.h): the patch geometry of Model.get_sky_patches and the neighbour lists of
    # Model.find_neighbors as the package's host logic computes them.  tools/reference_golden.jl ignores these arrays --
    # the reference builds its own.
    add("patch_box", [[p.bitmap_offset[0], p.bitmap_offset[1], p.active_pixel_bitmap.shape[0], p.active_pixel_bitmap.shape[1]]
                      for row in f.patches for p in row], np.int32)        # (S N) x 4, row s N + n: off_h, off_w, H2, W2
    add("patch_center", [[p.pixel_center[0], p.pixel_center[1]] for row in f.patches for p in row], np.float64)   # (S N) x 2
    off = np.cumsum([0] + [len(r) for r in f.neighbors])
    add("nbr_offsets", off, np.int64)                                      # S + 1 (CSR)
    add("nbr_index", [j for r in f.neighbors for j in r] or [0], np.int32)  # 0-based source ids (one dummy entry if none)
    os.makedirs(os.path.join(HERE, "raw"), exist_ok=True)
    off = 0
    with open(os.path.join(HERE, "raw", name + ".bin"), "wb") as fb, open(os.path.join(HERE, "raw", name + ".txt"), "w") as ft:
        for key, a, dt in arrays:
            data = np.asfortranarray(a.astype(dt)).tobytes(order="F")      # column-major
            ft.write("%s %s %d %s %d\n" % (key, np.dtype(dt).name, a.ndim, " ".join(str(d) for d in a.shape), off))
            fb.write(data)
            off += len(data)
    return off


if __name__ == "__main__":
    for nm in (sys.argv[1:] or DEFAULT):
        print(nm, export(nm), "bytes")
