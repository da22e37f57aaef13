"""Fitting the Gaussian-mixture PSF to a raw PSF stamp (input preparation, once per image).

  reference (src/PSF.jl)                                      here
  ----------------------------------------------------------  ---------------------------------------
  initialize_psf_params (:214-236), get_psf_transform bounds   _initial_params, _BOUNDS
      (:250-271)
  get_x_matrix_from_psf (:619-622)                             pixel offsets from the stamp centre
  evaluate_psf_fit! (:499-535): sum of squared residuals       _residuals
  fit_raw_psf_for_celeste (:635-673)                           fit_raw_psf_for_celeste
  trim_psf (:676-693)                                          trim_psf
  get_source_psf (:175-179)                                    get_source_psf

The reference minimises the squared error with its own Newton trust-region code over box-constrained parameters
(mean, axis ratio, angle, radius, weight per component); here the same objective, parametrisation, bounds and starting
point go through scipy's bounded trust-region least squares.  Same minimiser of the same function up to optimiser
tolerance; which of several equivalent optima (component order, angle modulo pi) comes out is not pinned.
"""
import math

import numpy as np

_BOUNDS = {"mu": (-5.0, 5.0), "axis_ratio": (0.1, 1.0), "angle": (-4 * math.pi, 4 * math.pi), "radius": (0.05, 10.0),
           "weight": (0.05, 2.0)}


def _initial_params(K: int) -> np.ndarray:
    """rows: (mu1, mu2, axis_ratio, angle, radius, weight)"""
    return np.array([[0.0, 0.0, 0.95, 0.0, math.sqrt(2 * (k + 1)), 1.0 / K] for k in range(K)])


def _bvn_cov(ab: float, angle: float, scale: float) -> np.ndarray:
    """get_bvn_cov (BivariateNormals.jl:29-43)"""
    cp, sp = math.cos(angle), math.sin(angle)
    R = np.array([[cp, -sp], [sp, cp]])
    return R @ np.diag([scale ** 2, (ab * scale) ** 2]) @ R.T


def _model(params: np.ndarray, x1: np.ndarray, x2: np.ndarray) -> np.ndarray:
    out = np.zeros_like(x1)
    for mu1, mu2, ab, ang, rad, w in params:
        S = _bvn_cov(ab, ang, rad)
        det = S[0, 0] * S[1, 1] - S[0, 1] ** 2
        d1, d2 = x1 - mu1, x2 - mu2
        q = (S[1, 1] * d1 * d1 - 2 * S[0, 1] * d1 * d2 + S[0, 0] * d2 * d2) / det
        out = out + w * np.exp(-0.5 * q) / (2 * math.pi * math.sqrt(det))
    return out


def fit_raw_psf_for_celeste(raw_psf: np.ndarray, K: int = 2, ftol: float = 1e-9):
    """Returns (psf [K x 6] = {alphaBar, xiBar1, xiBar2, tauBar11, tauBar12, tauBar22}, fitted parameter rows)."""
    from scipy.optimize import least_squares
    raw_psf = np.asarray(raw_psf, dtype=np.float64)
    c1, c2 = (raw_psf.shape[0] - 1) / 2 + 1, (raw_psf.shape[1] - 1) / 2 + 1
    x1 = (np.arange(1, raw_psf.shape[0] + 1) - c1)[:, None] * np.ones((1, raw_psf.shape[1]))
    x2 = np.ones((raw_psf.shape[0], 1)) * (np.arange(1, raw_psf.shape[1] + 1) - c2)[None, :]
    p0 = _initial_params(K)
    names = ("mu", "mu", "axis_ratio", "angle", "radius", "weight")
    lo = np.tile([_BOUNDS[n][0] for n in names], K)
    hi = np.tile([_BOUNDS[n][1] for n in names], K)

    def residuals(v):
        return (_model(v.reshape(K, 6), x1, x2) - raw_psf).ravel()
    res = least_squares(residuals, p0.ravel(), bounds=(lo, hi), method="trf", ftol=ftol, xtol=1e-12, gtol=1e-12,
                        max_nfev=2000)
    fit = res.x.reshape(K, 6)
    psf = np.zeros((K, 6))
    for k, (mu1, mu2, ab, ang, rad, w) in enumerate(fit):
        S = _bvn_cov(ab, ang, rad)
        psf[k] = [w, mu1, mu2, S[0, 0], S[0, 1], S[1, 1]]
    return psf, fit


def trim_psf(raw_psf: np.ndarray, trim_percent: float = 0.999) -> np.ndarray:
    """the smallest centred square holding trim_percent of the absolute mass"""
    h_mid, w_mid = -(-raw_psf.shape[0] // 2), -(-raw_psf.shape[1] // 2)   # cld(., 2), 1-based
    width = 1
    tot = np.abs(raw_psf).sum()

    def cut():
        return raw_psf[h_mid - 1 - width:h_mid + width, w_mid - 1 - width:w_mid + width]
    while np.abs(cut()).sum() < trim_percent * tot:
        width += 1
    return cut().copy()


def get_source_psf(world_loc, img, psf_K: int = 2) -> np.ndarray:
    """PSF.get_source_psf (PSF.jl:175-179): the mixture fitted to the PSF map at a world location"""
    pix = img.world_to_pix(world_loc)
    return fit_raw_psf_for_celeste(img.psfmap(pix[0], pix[1]), psf_K)[0]
