"""Output side of the path: optimised variational parameters -> catalog rows.

  reference (src/AccuracyBenchmark.jl)                         here
  ----------------------------------------------------------  ---------------------------------------
  canonical_angle (:162), color_from_fluxes (:150-156)         canonical_angle, color_from_fluxes
  fluxes_from_colors (:325-335), get_median_fluxes (:337-342)  fluxes_from_colors, get_median_fluxes
  variational_parameters_to_data_frame_row (:344-372)          variational_parameters_to_row
  celeste_to_df (:378-387)                                     celeste_to_rows
  flux_to_mag / mag_to_flux (:140-148), get_error_df,          flux_to_mag, mag_to_flux, get_error_row, is_good_row,
  is_good_row, get_scores_df, score_predictions (:795-977)     score_predictions
"""
import math
from typing import Dict, List, Optional, Sequence

import numpy as np

from .params import ids

COLUMNS = ["ra", "dec", "is_star", "gal_frac_dev", "gal_axis_ratio", "gal_radius_px", "gal_angle_deg", "flux_r_nmgy",
           "color_ug", "color_gr", "color_ri", "color_iz", "log_flux_r_stderr", "color_ug_stderr", "color_gr_stderr",
           "color_ri_stderr", "color_iz_stderr"]


def canonical_angle(gal_angle_deg: float) -> float:
    return gal_angle_deg - math.floor(gal_angle_deg / 180) * 180


def color_from_fluxes(flux1: float, flux2: float) -> Optional[float]:
    """log(flux2 / flux1); None (the reference's `missing`) for non-positive fluxes"""
    if flux1 <= 0 or flux2 <= 0:
        return None
    return math.log(flux2 / flux1)


def fluxes_from_colors(flux_r_nmgy: float, colors: Sequence[float]) -> np.ndarray:
    assert len(colors) == 4
    r = np.exp(np.asarray(colors, dtype=np.float64))
    f = np.zeros(5)
    f[2] = flux_r_nmgy
    f[3] = f[2] * r[2]
    f[4] = f[3] * r[3]
    f[1] = f[2] / r[1]
    f[0] = f[1] / r[0]
    return f


def get_median_fluxes(vs: np.ndarray, source_type: int) -> np.ndarray:
    """source_type 0 = star, 1 = galaxy (the reference's 1 / 2)"""
    return fluxes_from_colors(math.exp(vs[ids.flux_loc[source_type]]), vs[ids.color_mean[:, source_type]])


def variational_parameters_to_row(vs: np.ndarray) -> Dict[str, Optional[float]]:
    vs = np.asarray(vs, dtype=np.float64)
    row: Dict[str, Optional[float]] = {}
    row["ra"] = float(vs[ids.pos[0]])
    row["dec"] = float(vs[ids.pos[1]])
    row["is_star"] = float(vs[ids.is_star[0]])
    row["gal_frac_dev"] = float(vs[ids.gal_frac_dev])
    row["gal_axis_ratio"] = float(vs[ids.gal_axis_ratio])
    row["gal_radius_px"] = float(vs[ids.gal_radius_px] * math.sqrt(vs[ids.gal_axis_ratio]))
    row["gal_angle_deg"] = canonical_angle(180 / math.pi * float(vs[ids.gal_angle]))
    t = 0 if row["is_star"] > 0.5 else 1
    fl = get_median_fluxes(vs, t)
    row["flux_r_nmgy"] = float(fl[2])
    for name, a, b in (("color_ug", 0, 1), ("color_gr", 1, 2), ("color_ri", 2, 3), ("color_iz", 3, 4)):
        row[name] = color_from_fluxes(float(fl[a]), float(fl[b]))
    row["log_flux_r_stderr"] = math.sqrt(vs[ids.flux_scale[t]])
    for k, name in enumerate(("color_ug_stderr", "color_gr_stderr", "color_ri_stderr", "color_iz_stderr")):
        row[name] = math.sqrt(vs[ids.color_var[k, t]])
    return row


def celeste_to_rows(results) -> List[Dict[str, Optional[float]]]:
    """one row per OptimizedSource whose sky is not flagged bad"""
    return [variational_parameters_to_row(r.vs) for r in results if not r.is_sky_bad]


# ---- scoring against a ground-truth catalog (AccuracyBenchmark.jl:140-148, 795-931) -------------------------------

ASINH_SOFTENING_PARAMETERS = (1.4e-10, 0.9e-10, 1.2e-10, 1.8e-10, 7.4e-10)   # AccuracyBenchmark.jl, SDSS asinh magnitudes
COLOR_COLUMNS = ("color_ug", "color_gr", "color_ri", "color_iz")
ABSOLUTE_ERROR_COLUMNS = ("gal_frac_dev", "gal_axis_ratio", "gal_radius_px") + COLOR_COLUMNS


def flux_to_mag(flux_nmgy: float, band_index: int) -> float:
    """band_index 1..5 like the reference"""
    b = ASINH_SOFTENING_PARAMETERS[band_index - 1]
    return -2.5 / math.log(10) * (math.asinh(flux_nmgy * 1e-9 / (2 * b)) + math.log(b))


def mag_to_flux(mags: float, band_index: int) -> float:
    b = ASINH_SOFTENING_PARAMETERS[band_index - 1]
    return 1e9 * 2 * b * math.sinh(-math.log(10) / 2.5 * mags - math.log(b))


def degrees_to_diff(a: float, b: float) -> float:
    d = abs(a - b) % 180
    return min(d, 180 - d)


def catalog_entry_to_row(ce) -> Dict[str, Optional[float]]:
    """the ground-truth row of a CatalogEntry (the columns variational_parameters_to_row produces)"""
    fl = ce.star_fluxes if ce.is_star else ce.gal_fluxes
    row = {"ra": float(ce.pos[0]), "dec": float(ce.pos[1]), "is_star": 1.0 if ce.is_star else 0.0,
           "gal_frac_dev": ce.gal_frac_dev, "gal_axis_ratio": ce.gal_axis_ratio,
           "gal_radius_px": ce.gal_radius_px * math.sqrt(ce.gal_axis_ratio),
           "gal_angle_deg": canonical_angle(180 / math.pi * ce.gal_angle), "flux_r_nmgy": float(fl[2])}
    for k, name in enumerate(COLOR_COLUMNS):
        row[name] = color_from_fluxes(float(fl[k]), float(fl[k + 1]))
    return row


def get_error_row(truth: Dict, predicted: Dict) -> Dict[str, Optional[float]]:
    """get_error_df (:813-849) for one matched pair; positions are in pixels (world = pixel here)"""
    e: Dict[str, Optional[float]] = {}
    pg, tg = predicted["is_star"] < 0.5, truth["is_star"] < 0.5
    e["missed_stars"] = float(pg) if not tg else None
    e["missed_galaxies"] = float(not pg) if tg else None
    e["position"] = math.hypot(truth["ra"] - predicted["ra"], truth["dec"] - predicted["dec"])
    e["flux_r_mag"] = abs(flux_to_mag(truth["flux_r_nmgy"], 3) - flux_to_mag(predicted["flux_r_nmgy"], 3))
    e["flux_r_nmgy"] = abs(truth["flux_r_nmgy"] - predicted["flux_r_nmgy"])
    e["gal_angle_deg"] = degrees_to_diff(truth["gal_angle_deg"], predicted["gal_angle_deg"])
    for c in ABSOLUTE_ERROR_COLUMNS:
        e[c] = None if truth[c] is None or predicted[c] is None else abs(truth[c] - predicted[c])
    for c in COLOR_COLUMNS:
        if e[c] is not None:
            e[c] *= 2.5 / math.log(10)
    return e


def is_good_row(truth: Dict, error: Dict, column: str) -> bool:
    """:851-874"""
    if error[column] is None or math.isnan(error[column]):
        return False
    if truth["gal_radius_px"] is not None and truth["gal_radius_px"] > 20:
        return False
    if column in ("gal_axis_ratio", "gal_radius_px", "gal_angle_deg", "gal_frac_dev"):
        if truth["gal_frac_dev"] is not None and 0.05 < truth["gal_frac_dev"] < 0.95:
            return False
    if column == "gal_angle_deg" and truth["gal_axis_ratio"] is not None and truth["gal_axis_ratio"] > 0.6:
        return False
    return True


def score_predictions(truth_rows: List[Dict], predicted_rows: List[Dict]) -> Dict[str, Dict[str, float]]:
    """score_predictions / get_scores_df (:903-931, 967-977) for one set of predictions already matched row by row:
    {column: {"N": count of good rows, "first": mean error}}; columns with <= 1 good row are dropped."""
    assert len(truth_rows) == len(predicted_rows)
    errors = [get_error_row(t, p) for t, p in zip(truth_rows, predicted_rows)]
    out: Dict[str, Dict[str, float]] = {}
    for column in (errors[0].keys() if errors else []):
        good = [e[column] for t, e in zip(truth_rows, errors) if is_good_row(t, e, column)]
        if len(good) > 1:
            out[column] = {"N": len(good), "first": float(np.mean(good))}
    return out
