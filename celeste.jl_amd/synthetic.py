"""Synthetic SDSS-shaped fields: input preparation for tests and bench.

Recipe, following the reference's data-free construction paths:
  AccuracyBenchmark.make_image / make_template_images (AccuracyBenchmark.jl:553-570, 694-727):
      constant sky and nelec_per_nmgy per band, ConstantPSFMap(render_psf(psf, (51, 51)));
  test/SampleData.jl:30-34: identity WCS;
  Synthetic.gen_image! (Synthetic.jl:15-47): expected nmgy = sky + sum of star / galaxy light over
      radius-25 boxes (Model.write_star_nmgy!, write_galaxy_nmgy!, fsm_util.jl:349-400), times
      nelec_per_nmgy, Poisson sampled, stored Float32;
  AccuracyBenchmark.draw_source_params (AccuracyBenchmark.jl:400-452): catalog drawn from the prior.
The reference has no in-tree numeric defaults for sky, iota and PSF width; the values below are
builder-chosen, SDSS-like, and fixed.
"""
import json
import math
import os
from dataclasses import dataclass
from typing import List, Optional

import numpy as np

from . import cabi
from .model import (ConstantPSFMap, Image, ImagePatch, SDSSBackground, SDSSPSFMap, box_around_point, get_sky_patches,
                    make_psf, neighbor_map, render_psf)
from .params import CatalogEntry, catalog_init_source, perturb_params

SKY_NMGY = (0.20, 0.35, 0.60, 0.95, 1.40)
NELEC_PER_NMGY = (180.0, 750.0, 820.0, 600.0, 130.0)
PSF_ALPHA = (0.8, 0.2)
PSF_SIGMA = (1.25, 2.6)
PSF_BAND_SCALE = (1.10, 1.05, 1.00, 0.98, 1.00)
PRIOR_PROBABILITY_OF_STAR = 0.28

_HERE = os.path.dirname(os.path.abspath(__file__))


def load_prior() -> dict:
    with open(os.path.join(_HERE, "prior_tables.json")) as f:
        return json.load(f)


def galaxy_prototypes():
    """light_source_model.jl:45-75 -> (eta[2][8], nu[2][8])"""
    dev_amp = np.array([4.26347652e-2, 2.40127183e-1, 6.85907632e-1, 1.51937350,
                        2.83627243, 4.46467501, 5.72440830, 5.60989349])
    dev_var = np.array([2.23759216e-4, 1.00220099e-3, 4.18731126e-3, 1.69432589e-2,
                        6.84850479e-2, 2.87207080e-1, 1.33320254, 8.40215071])
    exp_amp = np.array([2.34853813e-3, 3.07995260e-2, 2.23364214e-1, 1.17949102, 4.33873750, 5.99820770])
    exp_var = np.array([1.20078965e-3, 8.84526493e-3, 3.91463084e-2, 1.39976817e-1, 4.60962500e-1, 1.50159566])
    eta = np.zeros((2, 8)); nu = np.zeros((2, 8))
    eta[0] = dev_amp / dev_amp.sum(); nu[0] = dev_var / 1.078031 ** 2
    eta[1, :6] = exp_amp / exp_amp.sum(); nu[1, :6] = exp_var / 0.928896 ** 2
    return eta, nu


def band_psf(b: int) -> np.ndarray:
    s = PSF_BAND_SCALE[b]
    return make_psf(PSF_ALPHA, [(0.0, 0.0)] * 2, [np.eye(2) * (sig * s) ** 2 for sig in PSF_SIGMA])


def blank_images(H: int, W: int) -> List[Image]:
    images = []
    for b in range(5):
        psf = band_psf(b)
        images.append(Image(pixels=np.zeros((H, W), dtype=np.float32), b=b + 1, psf=psf,
                            sky=np.full((H, W), SKY_NMGY[b], dtype=np.float32),
                            nelec_per_nmgy=np.full(H, NELEC_PER_NMGY[b], dtype=np.float32),
                            psfmap=ConstantPSFMap(render_psf(psf, (51, 51)))))
    return images


SDSS_GAIN = (1.62, 3.32, 4.71, 5.165, 4.745)   # electrons per DN (camcol 1 of the SDSS gain table, SDSSIO.jl:680-690)


def variable_images(H: int, W: int, seed: int = 0) -> List[Image]:
    """Images shaped like a real SDSS frame instead of the constant template: the three per-image inputs of the
    pixel term all vary (elbo_objective.jl:374-385: iota = img.nelec_per_nmgy[h] per ROW, img.sky[h, w] per PIXEL,
    and a per-patch star stamp, imaged_sources.jl:97-107):
      * sky = SDSSBackground.materialize(): bilinear interpolation of a small sky image (+-20 %, smooth) at per-row /
        per-column coordinates, times the per-row calibration (SDSSIO.jl:56-99);
      * nelec_per_nmgy[h] = gain / calibration[h] with a calibration drifting +-10 % along the rows (SDSSIO.jl:773);
      * psfmap = SDSSPSFMap with 3 eigen-images and first-order weight polynomials (SDSSIO.jl:239-299): eigen-image 0
        is the raster of the image-centre mixture `img.psf`, 1 and 2 widen / skew it away from the centre -- every patch
        gets its own stamp, hence its own spline, while `patch.psf` stays the image-centre fit (SURVEY.md trap A4).
    The objects are kept on the image (`background`, `psfmap`) so that fixtures can store their small defining arrays."""
    rng = np.random.Generator(np.random.PCG64([seed, 977]))
    images = []
    for b in range(5):
        psf = band_psf(b)
        # --- sky: a (9 x 7) sky image in DN, smooth, +-20 %
        nx, ny = 9, 7
        gx, gy = np.meshgrid(np.linspace(0, 1, nx), np.linspace(0, 1, ny), indexing="ij")
        ph = rng.uniform(0, 2 * np.pi, 3)
        rel = 1.0 + 0.2 * (0.6 * np.sin(2.1 * gx + ph[0]) * np.cos(1.7 * gy + ph[1]) + 0.4 * np.sin(3.0 * (gx + gy) + ph[2]))
        hh = np.arange(H) / max(H - 1, 1)
        drift = 1.0 + 0.1 * np.sin(2.6 * hh + rng.uniform(0, 2 * np.pi)) * np.cos(1.3 * hh + 0.4)
        calib0 = SDSS_GAIN[b] / NELEC_PER_NMGY[b]                      # nMgy per DN at drift = 1
        calibration = (calib0 * drift).astype(np.float32)
        sky_small = (SKY_NMGY[b] / calib0 * rel).astype(np.float32)    # DN
        # interpolation coordinates (1-based, with some constant extrapolation at both ends like real frames)
        sky_x = (np.linspace(0.6, nx + 0.4, H)).astype(np.float32)
        sky_y = (np.linspace(0.7, ny + 0.3, W)).astype(np.float32)
        bkg = SDSSBackground(sky_small, sky_x, sky_y, calibration)
        nelec = (np.float32(SDSS_GAIN[b]) / calibration).astype(np.float32)
        # --- PSF map: base raster + "wider" and "skewed" difference images, weights linear in (x, y), zero at the centre
        base = render_psf(psf, (51, 51))
        wide = render_psf(make_psf(PSF_ALPHA, [(0.0, 0.0)] * 2,
                                   [np.eye(2) * (1.25 * sig * PSF_BAND_SCALE[b]) ** 2 for sig in PSF_SIGMA]), (51, 51))
        skew = render_psf(make_psf(PSF_ALPHA, [(0.35, -0.25), (0.6, 0.4)],
                                   [np.array([[1.2, 0.3], [0.3, 0.9]]) * (sig * PSF_BAND_SCALE[b]) ** 2 for sig in PSF_SIGMA]),
                          (51, 51))
        rrows = np.stack([base.T.reshape(-1), (wide - base).T.reshape(-1), (skew - base).T.reshape(-1)], axis=1)
        RCS = 0.001
        cx, cy = RCS * (H / 2.0 - 1.0), RCS * (W / 2.0 - 1.0)
        a1, a2 = 0.45 / max(cx, 1e-9), 0.45 / max(cy, 1e-9)            # weights reach +-0.45 at the image edges
        cmat = np.zeros((2, 2, 3))
        cmat[0, 0, 0] = 1.0                                            # w0 = 1
        cmat[0, 0, 1], cmat[1, 0, 1] = -a1 * cx, a1                    # w1 = a1 (RCS (x - 1) - cx): 0 at the centre row
        cmat[0, 0, 2], cmat[0, 1, 2] = -a2 * cy, a2                    # w2 = a2 (RCS (y - 1) - cy): 0 at the centre column
        psfmap = SDSSPSFMap(rrows, 51, 51, cmat)
        img = Image(pixels=np.zeros((H, W), dtype=np.float32), b=b + 1, psf=psf, sky=bkg.materialize(),
                    nelec_per_nmgy=nelec, psfmap=psfmap)
        img.background = bkg
        images.append(img)
    return images


# ---- value-only light densities (numpy; used only to paint synthetic pixels) ----------------------
def _bspline_w(f):
    o = 1.0 - f
    return (o ** 3 / 6, 2.0 / 3 - f * f + f ** 3 / 2, 2.0 / 3 - o * o + o ** 3 / 2, f ** 3 / 6)


def star_density(coef: np.ndarray, xh: np.ndarray, xw: np.ndarray) -> np.ndarray:
    """softpluslikeinv(itp[xh, xw]) on arrays of 1-based stamp coordinates (fsm_util.jl:221-237)."""
    ix = np.clip(np.floor(xh).astype(int), 1, 50)
    iy = np.clip(np.floor(xw).astype(int), 1, 50)
    wx = _bspline_w(xh - ix)
    wy = _bspline_w(xw - iy)
    y = np.zeros(np.broadcast(xh, xw).shape)
    for a in range(4):
        for b in range(4):
            y = y + coef[ix - 1 + a, iy - 1 + b] * wx[a] * wy[b]
    return np.where(y < 0, 1e-3 * np.exp(np.minimum(y, 0)), 1e-3 * (y + 1))


def galaxy_density(psf: np.ndarray, m_pos, frac_dev, axis_ratio, angle, radius, hh, ww) -> np.ndarray:
    """sum over PSF (x) prototype components (fsm_util.jl:37-65, 194-219), value only."""
    eta, nu = galaxy_prototypes()
    cp, sp = math.cos(angle), math.sin(angle)
    ab = axis_ratio ** 2 - 1
    s2 = radius ** 2
    x11 = s2 * (1 + ab * sp * sp); x22 = s2 * (1 + ab * cp * cp); x12 = -s2 * cp * sp * ab
    out = np.zeros(np.broadcast(hh, ww).shape)
    for i in range(2):
        th = frac_dev if i == 0 else 1.0 - frac_dev
        for j in range(8 if i == 0 else 6):
            for a, xi1, xi2, t11, t12, t22 in psf:
                s11, s12, s22 = t11 + nu[i, j] * x11, t12 + nu[i, j] * x12, t22 + nu[i, j] * x22
                det = s11 * s22 - s12 * s12
                d1 = hh - (xi1 + m_pos[0]); d2 = ww - (xi2 + m_pos[1])
                q = (s22 * d1 * d1 - 2 * s12 * d1 * d2 + s11 * d2 * d2) / det
                out = out + th * a * eta[i, j] / (2 * math.pi * math.sqrt(det)) * np.exp(-0.5 * q)
    return out


def render_expected_image(img: Image, catalog: List[CatalogEntry]) -> np.ndarray:
    """Synthetic.gen_image! with expectation=true for one image, before the nelec scaling: sky + sources, in nmgy."""
    nm = img.sky.astype(np.float64).copy()
    constant_map = isinstance(img.psfmap, ConstantPSFMap)
    coef = cabi.spline_prefilter(img.psfmap(0, 0)) if constant_map else None
    pc = (np.array([ce.pos for ce in catalog], dtype=float).reshape(-1, 2) - img.wcs_world0) @ img.wcs_jacobian.T \
        + img.wcs_pix0
    near = np.flatnonzero((pc[:, 0] > -27) & (pc[:, 0] < img.H + 28) & (pc[:, 1] > -27) & (pc[:, 1] < img.W + 28))
    for ce in (catalog[s] for s in near):   # the others' radius-25 boxes miss the image
        box = box_around_point(img, ce.pos, 25)
        p = ImagePatch.from_box(img, box)
        (h0, h1), (w0, w1) = p.box
        if h1 < h0 or w1 < w0:
            continue
        hh = np.arange(h0, h1 + 1, dtype=float)[:, None]
        ww = np.arange(w0, w1 + 1, dtype=float)[None, :]
        m = p.wcs_jacobian @ (np.asarray(ce.pos, float) - p.world_center) + p.pixel_center
        if ce.is_star:
            if not constant_map:   # the stamp of the source's own patch (its box centre), as the model will see it
                coef = cabi.spline_prefilter(img.psfmap(p.pixel_center[0], p.pixel_center[1]))
            dens = star_density(coef, hh - m[0] + 26, ww - m[1] + 26) * ce.star_fluxes[img.b - 1]
        else:
            dens = galaxy_density(img.psf, m, ce.gal_frac_dev, ce.gal_axis_ratio, ce.gal_angle,
                                  ce.gal_radius_px, hh, ww) * ce.gal_fluxes[img.b - 1]
        nm[h0 - 1:h1, w0 - 1:w1] += dens
    return nm


def render_expected_nmgy(images: List[Image], catalog: List[CatalogEntry]) -> List[np.ndarray]:
    return [render_expected_image(img, catalog) for img in images]


def _sample_image(img: Image, catalog, seed) -> np.ndarray:
    el = render_expected_image(img, catalog) * img.nelec_per_nmgy.astype(np.float64)[:, None]
    return np.random.Generator(np.random.PCG64(seed)).poisson(el).astype(np.float32)


_POOL_ARGS = None


def _sample_image_job(n):
    images, catalog, seed = _POOL_ARGS
    return _sample_image(images[n], catalog, [seed, n])


def gen_images(images: List[Image], catalog: List[CatalogEntry], rng: np.random.Generator,
               expectation: bool = False, workers: int = 1, seed: int = 0) -> None:
    """Synthetic.gen_images! (Synthetic.jl:30-58).  workers > 1 (many-image problems): the images are rendered by
    forked worker processes and image n is Poisson-sampled from its own generator PCG64([seed, n]) instead of the
    shared `rng` (the same pixels for any number of workers > 1)."""
    if workers > 1 and not expectation:
        import multiprocessing as mp
        global _POOL_ARGS
        _POOL_ARGS = (images, catalog, seed)
        try:
            pool = mp.get_context("fork").Pool(workers)
            try:
                for img, px in zip(images, pool.map(_sample_image_job, range(len(images)), chunksize=1)):
                    img.pixels = px
            finally:
                pool.close()    # let the workers run out and exit by themselves (no SIGTERM: tools that wrap the
                pool.join()     # process, e.g. profilers, may intercept it)
        finally:
            _POOL_ARGS = None
        return
    for img in images:
        el = render_expected_image(img, catalog) * img.nelec_per_nmgy.astype(np.float64)[:, None]
        if not expectation:
            el = rng.poisson(el).astype(np.float64)
        img.pixels = el.astype(np.float32)


# ---- catalogs -----------------------------------------------------------------------------------
def fluxes_from_colors(r_flux: float, colors) -> np.ndarray:
    """Synthetic.sample_fluxes (Synthetic.jl:66-77)"""
    l = np.zeros(5)
    l[2] = r_flux
    l[3] = l[2] * math.exp(colors[2])
    l[4] = l[3] * math.exp(colors[3])
    l[1] = l[2] / math.exp(colors[1])
    l[0] = l[1] / math.exp(colors[0])
    return l


def draw_source(prior: dict, rng: np.random.Generator, pos, force_star: Optional[bool] = None) -> CatalogEntry:
    """AccuracyBenchmark.draw_source_params (AccuracyBenchmark.jl:400-452)"""
    is_star = bool(rng.random() < PRIOR_PROBABILITY_OF_STAR) if force_star is None else force_star
    i = 0 if is_star else 1
    flux_r = math.exp(rng.normal(prior["flux_mean"][i], math.sqrt(prior["flux_var"][i])))
    d = rng.choice(8, p=np.asarray(prior["k"][i]) / np.sum(prior["k"][i]))
    cov = np.asarray(prior["color_cov"][i][d]).reshape(4, 4)
    colors = rng.multivariate_normal(np.asarray(prior["color_mean"][i][d]), cov)
    fl = fluxes_from_colors(flux_r, colors)
    if is_star:
        return CatalogEntry(np.asarray(pos, float), True, fl, fl.copy(), 0.5, 0.8, 0.0, 0.2)
    radius = math.exp(rng.normal(prior["gal_radius_px_mean"], math.sqrt(prior["gal_radius_px_var"])))
    angle = math.radians(rng.uniform(0, 180))
    axis_ratio = rng.beta(2, 2)
    frac_dev = rng.beta(0.5, 0.5)
    return CatalogEntry(np.asarray(pos, float), False, fl.copy(), fl, float(frac_dev), float(axis_ratio),
                        float(angle), float(radius))


@dataclass
class Field:
    images: List[Image]
    catalog: List[CatalogEntry]
    patches: List[List[ImagePatch]]
    neighbors: List[List[int]]
    vp: np.ndarray          # S x 44
    name: str = ""


def make_field(H: int, W: int, n_sources: int, seed: int, stars_only: bool = False, perturb: bool = True,
               nan_fraction: float = 0.0, margin: int = 26, name: str = "", variable: bool = False) -> Field:
    """Configs 2 / 3 of SURVEY.md 8(d): uniform positions with a margin, prior-drawn sources.
    variable=True: SDSS-like varying sky plane, per-row calibration and per-patch PSF stamps (`variable_images`)
    instead of the constant template of AccuracyBenchmark.make_image."""
    rng = np.random.Generator(np.random.PCG64(seed))
    prior = load_prior()
    images = variable_images(H, W, seed) if variable else blank_images(H, W)
    catalog = []
    for _ in range(n_sources):
        pos = (rng.uniform(margin, H - margin), rng.uniform(margin, W - margin))
        catalog.append(draw_source(prior, rng, pos, force_star=True if stars_only else None))
    gen_images(images, catalog, rng)
    if nan_fraction > 0:
        for img in images:
            mask = rng.random(img.pixels.shape) < nan_fraction
            img.pixels[mask] = np.nan
    patches = get_sky_patches(images, catalog)
    nbrs = neighbor_map(patches)
    vp = [catalog_init_source(ce) for ce in catalog]
    if perturb:
        perturb_params(vp)
    return Field(images, catalog, patches, nbrs, np.stack(vp), name)


def make_multifield(grid=(2, 2), H: int = 256, W: int = 256, overlap: float = 0.10, n_sources: int = 120,
                    seed: int = 5, perturb: bool = True, margin: int = 8, sparse: bool = False,
                    workers: int = 1) -> Field:
    """Config 5 of SURVEY.md 8(d) in miniature: a grid of overlapping fields (5 bands each) on one world
    coordinate system (world = global pixel coordinates; each image has its own affine offset).  A source
    has non-empty patches only in the images it overlaps; the others are the reference's empty boxes
    (clamp_box, imaged_sources.jl:10-14)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    prior = load_prior()
    images: List[Image] = []
    step_h, step_w = int(round(H * (1 - overlap))), int(round(W * (1 - overlap)))
    for gi in range(grid[0]):
        for gj in range(grid[1]):
            for im in blank_images(H, W):
                im.wcs_world0 = np.array([float(gi * step_h), float(gj * step_w)])
                images.append(im)
    tot_h, tot_w = step_h * (grid[0] - 1) + H, step_w * (grid[1] - 1) + W
    catalog = []
    for _ in range(n_sources):
        pos = (rng.uniform(margin, tot_h - margin), rng.uniform(margin, tot_w - margin))
        catalog.append(draw_source(prior, rng, pos))
    gen_images(images, catalog, rng, workers=workers, seed=seed)
    patches = get_sky_patches(images, catalog, sparse=sparse)
    nbrs = neighbor_map(patches)
    vp = [catalog_init_source(ce) for ce in catalog]
    if perturb:
        perturb_params(vp)
    return Field(images, catalog, patches, nbrs, np.stack(vp), "multifield_%dx%d" % grid)


SAMPLE_STAR_FLUXES = np.array([4.451805E+03, 1.491065E+03, 2.264545E+03, 2.027004E+03, 1.846822E+04])
SAMPLE_GALAXY_FLUXES = np.array([1.377666E+01, 5.635334E+01, 1.258656E+02, 1.884264E+02, 2.351820E+02]) * 100


def sample_ce(pos, is_star: bool) -> CatalogEntry:
    """test/SampleData.jl:120-123"""
    return CatalogEntry(np.asarray(pos, float), is_star, SAMPLE_STAR_FLUXES.copy(), SAMPLE_GALAXY_FLUXES.copy(),
                        0.1, 0.7, math.pi / 4, 4.0)


def make_sample_dataset(kind: str = "star", seed: int = 1, perturb: bool = True) -> Field:
    """gen_sample_star_dataset / gen_sample_galaxy_dataset / gen_two_body_dataset /
    gen_three_body_dataset analogues (test/SampleData.jl:161-236) on synthetic image metadata."""
    rng = np.random.Generator(np.random.PCG64(seed))
    if kind == "star":
        H, W, catalog = 20, 23, [sample_ce([10.1, 12.2], True)]
    elif kind == "galaxy":
        H, W, catalog = 20, 23, [sample_ce([8.5, 9.6], False)]
    elif kind == "two_body":
        H, W, catalog = 20, 23, [sample_ce([4.5, 3.6], False), sample_ce([10.1, 12.1], True)]
    elif kind == "three_body":
        H, W, catalog = 112, 238, [sample_ce([4.5, 3.6], False), sample_ce([60.1, 82.2], True),
                                   sample_ce([71.3, 100.4], False)]
    else:
        raise ValueError(kind)
    images = blank_images(H, W)
    gen_images(images, catalog, rng)
    patches = get_sky_patches(images, catalog)
    nbrs = neighbor_map(patches)
    vp = [catalog_init_source(ce) for ce in catalog]
    if perturb:
        perturb_params(vp)
    return Field(images, catalog, patches, nbrs, np.stack(vp), kind)
