"""Output side of the path: optimised variational parameters -> catalog rows.

  reference (src/AccuracyBenchmark.jl)                         here
  ----------------------------------------------------------  ---------------------------------------
  canonical_angle (:162), color_from_fluxes (:150-156)         canonical_angle, color_from_fluxes
  fluxes_from_colors (:325-335), get_median_fluxes (:337-342)  fluxes_from_colors, get_median_fluxes
  variational_parameters_to_data_frame_row (:344-372)          variational_parameters_to_row
  celeste_to_df (:378-387)                                     celeste_to_rows
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
